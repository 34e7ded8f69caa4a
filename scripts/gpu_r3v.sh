#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7,8,9,10,11
python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "base:" "p5k:STMPC_PRIO=5000" "p10k:STMPC_PRIO=10000" "p15k:STMPC_PRIO=15000" "p20k:STMPC_PRIO=20000" "p30k:STMPC_PRIO=30000" "p45k:STMPC_PRIO=45000" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | grep "median\|DIFFER"
