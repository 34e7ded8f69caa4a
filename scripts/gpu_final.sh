#!/bin/bash
# what the driver runs at round end: GPU suite, smoke, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -2
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; echo "bench rc=$?"; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print(d["metric"], d["value"], d["ms_per_step"], d["value_seed_median"], d["parity_vs_oracle"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
