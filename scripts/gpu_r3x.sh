#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3x; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7,8,9,10,11
python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "base:" "heavy:STMPC_HEAVY_FIRST=1" "heavy_p25k:STMPC_HEAVY_FIRST=1;STMPC_PRIO=25000" "heavy_p40k:STMPC_HEAVY_FIRST=1;STMPC_PRIO=40000" "cap375:STMPC_BAND_CAP=375" "cap525:STMPC_BAND_CAP=525" "tube64:STMPC_TUBE=64" "tube128:STMPC_TUBE=128" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | grep "median\|DIFFER"
python scripts/lab/sweep.py $O/s8k.json 8192 1000,1,2,3 "base8k:" "heavy8k:STMPC_HEAVY_FIRST=1" 2>&1 | grep "median\|DIFFER" | tee -a $O/sweep.log
python scripts/lab/sweep.py $O/s16k.json 16384 1000,1 "base16k:" 2>&1 | grep "median\|DIFFER" | tee -a $O/sweep.log
