#!/bin/bash
# with the dense bounding passes a wider tube / a higher node cap cost less than before: re-sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3ah; mkdir -p $O
SEEDS=1000,1,2,3,4,5
timeout 500 python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "base:" "t112:STMPC_TUBE=112" "t127:STMPC_TUBE=127" "cap550:STMPC_BAND_CAP=550" "cap650:STMPC_BAND_CAP=650" "t127cap550:STMPC_TUBE=127;STMPC_BAND_CAP=550" "prio24k:STMPC_PRIO=24000" "prio40k:STMPC_PRIO=40000" 2>&1 < /dev/null | grep -v amdgpu.ids > $O/sweep.log
grep "median\|DIFFER" $O/sweep.log; grep "seed  1000" $O/sweep.log
