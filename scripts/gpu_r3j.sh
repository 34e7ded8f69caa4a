#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "golden or bounded or oracle_seeded or overflow or config4 or candidate or general or other_parameter" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
SEEDS=1000,1,2,3,4,5,6,7
timeout 900 python scripts/lab/sweep.py $O/sweep.json 4096 $SEEDS "tube96:" "notube:STMPC_TUBE=0" "tube48:STMPC_TUBE=48" "tube64:STMPC_TUBE=64" "tube128:STMPC_TUBE=128" "tube192:STMPC_TUBE=192" "tube64c600:STMPC_TUBE=64;STMPC_BAND_CAP=600" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
timeout 300 python scripts/lab/sweep.py $O/sweep16k.json 16384 1000 "tube96_16k:" "notube_16k:STMPC_TUBE=0" "tube64_16k:STMPC_TUBE=64" 2>&1 | grep -v amdgpu.ids | tee $O/sweep16k.log
