#!/bin/bash
# full GPU suite, then the bench lines of every workload (short)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/full; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 200 python bench.py --no-cpu-baseline > $O/bench_h40.json 2> $O/bench_h40.err < /dev/null; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_h40.json").read().strip().splitlines()[-1]); print("h40a21", d["value"], d.get("value_seed_median"), d["ms_per_step"])
except Exception as ex: print("bench parse failed", ex)
PY
timeout 200 python bench.py --workload default --no-cpu-baseline --seeds= > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -c 400 $O/bench_default.json | head -c 400; echo
