#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "r105:STMPC_RETRY=1.05,1.3,4" "r105m2:STMPC_RETRY=1.05,1.3,4;STMPC_RETRY_MOVE=2" "r103:STMPC_RETRY=1.03,1.3,4" "r110:STMPC_RETRY=1.1,1.5,4" "r108:STMPC_RETRY=1.08,1.4,4" "r105m1:STMPC_RETRY=1.05,1.3,4;STMPC_RETRY_MOVE=1" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | grep "median\|DIFFER"
