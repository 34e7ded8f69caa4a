#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "golden or bounded or oracle_seeded or overflow or config4 or candidate or general" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
SEEDS=1000,1,2,3,4,5,6,7
timeout 900 python scripts/lab/sweep.py $O/sweep.json 4096 $SEEDS "repair:" "norepair:STMPC_REPAIR=0" "rc800:STMPC_REPAIR_CAP=800" "rc2000:STMPC_REPAIR_CAP=2000" "rb1:STMPC_REPAIR_BAND=1" "rb4:STMPC_REPAIR_BAND=4" "rc1200c300:STMPC_BAND_CAP=300" "rc1200c600:STMPC_BAND_CAP=600" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
timeout 300 python scripts/lab/sweep.py $O/sweep8k.json 8192 1000,1,2 "repair8k:" "norepair8k:STMPC_REPAIR=0" 2>&1 | grep -v amdgpu.ids | tee $O/sweep8k.log
timeout 300 python scripts/lab/sweep.py $O/sweep16k.json 16384 1000 "repair16k:" "norepair16k:STMPC_REPAIR=0" 2>&1 | grep -v amdgpu.ids | tee $O/sweep16k.log
