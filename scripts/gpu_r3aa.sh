#!/bin/bash
# bound inflation sweep: fewer repeated exact passes against more exact nodes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3aa; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
timeout 400 python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "base:" "i0001:STMPC_BOUND_INFL=1.0001" "i0005:STMPC_BOUND_INFL=1.0005" "i001:STMPC_BOUND_INFL=1.001" "i003:STMPC_BOUND_INFL=1.003" "i01:STMPC_BOUND_INFL=1.01" "i03:STMPC_BOUND_INFL=1.03" "i05:STMPC_BOUND_INFL=1.05" 2>&1 | grep -v amdgpu.ids > $O/sweep.log
grep "median\|DIFFER" $O/sweep.log; grep "seed  1000" $O/sweep.log
