#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
SEEDS=1000,1,2,3,4
specs=("base:")
for k in 32 64 96; do for at in 50 75 100; do specs+=("k${k}a${at}:STMPC_RETIRE_CUS=$k;STMPC_RETIRE_AT=$at"); done; done
specs+=("k64a75c600:STMPC_RETIRE_CUS=64;STMPC_RETIRE_AT=75;STMPC_BAND_CAP=600" "k48a60:STMPC_RETIRE_CUS=48;STMPC_RETIRE_AT=60" "k128a50:STMPC_RETIRE_CUS=128;STMPC_RETIRE_AT=50")
specs+=("t3a:STMPC_TIERS=2048,4096,8192;STMPC_NW=4,8,8;STMPC_PEN_CELLS=1024,2048,4096" "t3b:STMPC_TIERS=2048,4096,8192;STMPC_NW=4,4,8;STMPC_PEN_CELLS=1024,2048,4096" "t2_4096:STMPC_TIERS=2048,4096;STMPC_NW=4,8;STMPC_PEN_CELLS=1024,2048")
timeout 900 python scripts/lab/sweep.py $O/sweep.json 4096 $SEEDS "${specs[@]}" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
timeout 300 python scripts/lab/sweep.py $O/sweep8k.json 8192 1000,1,2 "base8k:" "k64a75_8k:STMPC_RETIRE_CUS=64;STMPC_RETIRE_AT=75" "k32a50_8k:STMPC_RETIRE_CUS=32;STMPC_RETIRE_AT=50" 2>&1 | grep -v amdgpu.ids | tee $O/sweep8k.log
