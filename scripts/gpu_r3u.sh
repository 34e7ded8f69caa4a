#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7,8,9,10,11
specs=("base:")
for k in 16 32 48; do for at in 40 60 80; do specs+=("k${k}a${at}:STMPC_RETIRE_CUS=$k;STMPC_RETIRE_AT=$at"); done; done
specs+=("heavy:STMPC_HEAVY_FIRST=1" "heavy_k32a60:STMPC_HEAVY_FIRST=1;STMPC_RETIRE_CUS=32;STMPC_RETIRE_AT=60")
python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "${specs[@]}" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | grep "median\|DIFFER"
