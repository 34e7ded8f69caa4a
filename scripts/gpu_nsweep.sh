#!/bin/bash
# solves/s against states per launch (library events, seed-median of 3 seeds)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/nsweep; mkdir -p $O
for n in 512 1024 2048 4096 8192 16384 32768; do
  timeout 200 python scripts/lab/sweep.py $O/s_$n.json $n 1000,1,2 "n$n:" 2>&1 < /dev/null | grep "median"
done
