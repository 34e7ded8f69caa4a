#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
timeout 900 python scripts/lab/sweep.py $O/sweep.json 4096 $SEEDS "base:" "band1200:STMPC_BAND=1200" "band2700:STMPC_BAND=2700" "band3600:STMPC_BAND=3600" "b2m2:STMPC_BAND2_MULT=2" "b2m6:STMPC_BAND2_MULT=6" \
  "gsh3:STMPC_GSH=3" "gsh2:STMPC_GSH=2" "nosplit:STMPC_SPLIT=0" "heavy:STMPC_HEAVY_FIRST=1" "wpc12:STMPC_WAVES_PER_CU=12" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
