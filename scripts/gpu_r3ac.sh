#!/bin/bash
# tube half-width of the guided bounding attempt: quality of the bound (exact nodes, repeated passes, overflows)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3ac; mkdir -p $O
SEEDS=1000,1,2,3
timeout 400 python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "t96:" "t64:STMPC_TUBE=64" "t48:STMPC_TUBE=48" "t31:STMPC_TUBE=31" "t24:STMPC_TUBE=24" "t16:STMPC_TUBE=16" "t8:STMPC_TUBE=8" 2>&1 | grep -v amdgpu.ids > $O/sweep.log
grep "median\|DIFFER" $O/sweep.log; grep "seed  1000" $O/sweep.log
