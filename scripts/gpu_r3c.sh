#!/bin/bash
# GPU run r3c: draining whole compute units (STMPC_RETIRE), a 4096-cell middle window
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
SEEDS=1000,1,2,3,4
timeout 600 python scripts/lab/sweep.py $O/sweep_ret.json 4096 $SEEDS "base:" "ret8:STMPC_RETIRE=8" "ret16:STMPC_RETIRE=16" "ret24:STMPC_RETIRE=24" "ret32:STMPC_RETIRE=32" "ret48:STMPC_RETIRE=48" "ret64:STMPC_RETIRE=64" \
  "ret16c600:STMPC_RETIRE=16,STMPC_BAND_CAP=600" "ret32c600:STMPC_RETIRE=32,STMPC_BAND_CAP=600" \
  "t3a:STMPC_TIERS=2048,4096,8192,STMPC_NW=4,8,8,STMPC_PEN_CELLS=1024,2048,4096" "t3b:STMPC_TIERS=2048,4096,8192,STMPC_NW=4,4,8,STMPC_PEN_CELLS=1024,2048,4096" \
  "t3a_ret16:STMPC_RETIRE=16,STMPC_TIERS=2048,4096,8192,STMPC_NW=4,8,8,STMPC_PEN_CELLS=1024,2048,4096" 2>&1 | grep -v amdgpu.ids | tee $O/sweep_ret.log
timeout 300 python scripts/lab/sweep.py $O/sweep_ret8k.json 8192 1000,1,2 "base8k:" "ret16_8k:STMPC_RETIRE=16" "ret32_8k:STMPC_RETIRE=32" 2>&1 | grep -v amdgpu.ids | tee $O/sweep_ret8k.log
python -m pytest tests -m gpu -x -q -k "golden or bounded or oracle_seeded" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
STMPC_RETIRE=16 python -m pytest tests -m gpu -x -q -k "golden or bounded or oracle_seeded or config4" > $O/pytest_gpu_ret.log 2>&1; echo "pytest(retire) rc=$?"; tail -3 $O/pytest_gpu_ret.log
