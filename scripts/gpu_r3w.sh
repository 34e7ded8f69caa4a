#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3w; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7,8,9,10,11
specs=("base:" "p25k:STMPC_PRIO=25000" "p30k:STMPC_PRIO=30000" "p35k:STMPC_PRIO=35000")
for t in 10000 14000 18000; do specs+=("m1_$t:STMPC_PRIO_MODE=1;STMPC_PRIO=$t"); done
for t in 20000 30000 40000; do specs+=("m2_$t:STMPC_PRIO_MODE=2;STMPC_PRIO=$t"); done
for t in 20000 35000 50000; do specs+=("m3_$t:STMPC_PRIO_MODE=3;STMPC_PRIO=$t"); done
python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "${specs[@]}" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | grep "median\|DIFFER"
