#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
timeout 900 python scripts/lab/sweep.py $O/sweep.json 4096 $SEEDS "base:" "heavy:STMPC_HEAVY_FIRST=1" "heavy_k32a60:STMPC_HEAVY_FIRST=1;STMPC_RETIRE_CUS=32;STMPC_RETIRE_AT=60" "heavy_k48a50:STMPC_HEAVY_FIRST=1;STMPC_RETIRE_CUS=48;STMPC_RETIRE_AT=50" "heavy_k64a40:STMPC_HEAVY_FIRST=1;STMPC_RETIRE_CUS=64;STMPC_RETIRE_AT=40" "heavy_k32a40:STMPC_HEAVY_FIRST=1;STMPC_RETIRE_CUS=32;STMPC_RETIRE_AT=40" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
timeout 300 python scripts/lab/sweep.py $O/sweep8k.json 8192 1000,1,2 "base8k:" "heavy8k:STMPC_HEAVY_FIRST=1" 2>&1 | grep -v amdgpu.ids | tee $O/sweep8k.log
