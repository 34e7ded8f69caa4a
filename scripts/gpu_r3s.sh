#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3s; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7,8,9,10,11
python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "base:" "move1:STMPC_RETRY_MOVE=1" "move2:STMPC_RETRY_MOVE=2" "r13:STMPC_RETRY=1.02,1.3,4" "r13move2:STMPC_RETRY=1.02,1.3,4;STMPC_RETRY_MOVE=2" "r105:STMPC_RETRY=1.05,1.3,4" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | grep "median\|DIFFER"
