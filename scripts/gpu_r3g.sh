#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
python bench.py --workload episodes --steps 50 --warmup 5 > $O/bench_episodes.json 2> $O/bench_episodes.err; tail -c 1200 $O/bench_episodes.json; tail -3 $O/bench_episodes.err
python bench.py --workload combined --steps 10 --warmup 2 > $O/bench_combined.json 2> $O/bench_combined.err; tail -c 600 $O/bench_combined.json
python bench.py --steps 20 --warmup 5 > $O/bench_h40a21.json 2> $O/bench_h40a21.err; tail -c 3000 $O/bench_h40a21.json
