#!/bin/bash
# second window: dense bound repair before a queued search goes on (STMPC_REPAIR_CAP)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3aj; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
timeout 500 python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "base:" "rep450:STMPC_REPAIR_CAP=450" "rep700:STMPC_REPAIR_CAP=700" "rep1000:STMPC_REPAIR_CAP=1000" "rep1400:STMPC_REPAIR_CAP=1400" "rep2000:STMPC_REPAIR_CAP=2000" 2>&1 < /dev/null | grep -v amdgpu.ids > $O/sweep.log
grep "median\|DIFFER" $O/sweep.log; grep "seed  1000" $O/sweep.log
