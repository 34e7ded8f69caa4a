#!/bin/bash
# second window tightens the bound of a queued episode before it goes on (STMPC_REPAIR_CAP)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3ab; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
timeout 400 python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "base:" "rep600:STMPC_REPAIR_CAP=600" "rep1200:STMPC_REPAIR_CAP=1200" "rep2400:STMPC_REPAIR_CAP=2400" 2>&1 | grep -v amdgpu.ids > $O/sweep.log
grep "median\|DIFFER" $O/sweep.log; grep "seed  1000" $O/sweep.log
