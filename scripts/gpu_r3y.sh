#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
specs=("base:")
for nd in 200 400 700; do for b in 64 192; do specs+=("m${nd}b${b}:STMPC_MIGRATE=$nd;STMPC_MIGRATE_BUDGET=$b"); done; done
timeout 150 python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "${specs[@]}" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | grep "median\|DIFFER"
