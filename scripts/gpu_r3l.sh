#!/bin/bash
# round-3 refresh: full GPU suite, profiles (kernel trace + counter passes + phase shares), bench lines of every workload
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3l; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 400 bash scripts/profile_gpu.sh r3 > $O/prof_r3.log 2>&1 < /dev/null
timeout 300 bash scripts/profile_gpu.sh r3_default --workload default > $O/prof_r3_default.log 2>&1 < /dev/null
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench_h40a21.json 2> $O/bench_h40a21.err < /dev/null; tail -c 400 $O/bench_h40a21.json; echo
timeout 100 python bench.py --steps 20 --warmup 5 --episodes 8192 --no-cpu-baseline > $O/bench_h40a21_n8k.json 2>/dev/null < /dev/null
timeout 100 python bench.py --steps 10 --warmup 3 --episodes 16384 --no-cpu-baseline > $O/bench_h40a21_n16k.json 2>/dev/null < /dev/null
timeout 200 python bench.py --steps 20 --warmup 5 --workload default > $O/bench_default.json 2>/dev/null < /dev/null
timeout 200 python bench.py --steps 20 --warmup 5 --workload control > $O/bench_control.json 2>/dev/null < /dev/null
timeout 200 python bench.py --steps 10 --warmup 3 --workload combined > $O/bench_combined.json 2>/dev/null < /dev/null
timeout 200 python bench.py --steps 50 --warmup 5 --workload episodes > $O/bench_episodes.json 2>/dev/null < /dev/null
for f in $O/bench_*.json; do [ -s "$f" ] && python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); print("$f", round(d["value"]), d["unit"], round(d["ms_per_step"],3), d.get("value_seed_median"), d.get("parity_vs_oracle"))
PY
done
