#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3r; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
python scripts/lab/sweep.py $O/s0.json 4096 $SEEDS "f16:" 2>&1 | grep "median\|DIFFER" | tee $O/sweep.log
for v in f24 f24u8 f16u8; do STMPC_LIB=$PWD/variants/libstmpc_$v.so python scripts/lab/sweep.py $O/s_$v.json 4096 $SEEDS "$v:" 2>&1 | grep "median\|DIFFER" | tee -a $O/sweep.log; done
