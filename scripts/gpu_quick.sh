#!/bin/bash
# usage: gpu_quick.sh "<tag> ENV=.. ENV=.." ...   -- pytest -m gpu, then one bench line per argument
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
if [ "$SKIP_TESTS" != "1" ]; then python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log; fi
for spec in "$@"; do
  set -- $spec; tag=$1; shift
  env "$@" python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1]); print("$tag", round(d["value"]), round(d["ms_per_step"],3), d["tiers"])
except Exception as e: print("$tag FAILED", e); print(open("$O/bench_$tag.err").read()[-800:])
PY
done
