#!/bin/bash
# kernel trace of a few solves: per-kernel durations.  usage: gpu_kt.sh [ENV=V;ENV=V]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/kt; mkdir -p $O; rm -rf $O/*
timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python $GRAFT_REPO_ROOT/scripts/lab/sweep.py $O/s.json 4096 1000 "x:$1" > $O/log.txt 2>&1 < /dev/null
f=$(ls $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -12 "$f" | cut -c1-200; else echo "no stats file"; tail -5 $O/log.txt; fi
