#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
python scripts/lab/sweep.py $O/sweep_main.json 4096 $SEEDS "ilp:" "ilp_nw16:STMPC_NW=4,16" "ilp_nw12:STMPC_NW=4,12" 2>&1 | grep median | tee $O/sweep.log
STMPC_LIB=$PWD/variants/libstmpc_ub8.so python scripts/lab/sweep.py $O/sweep_ub8.json 4096 $SEEDS "ub8:" "ub8_nw16:STMPC_NW=4,16" 2>&1 | grep median | tee -a $O/sweep.log
STMPC_LIB=$PWD/variants/libstmpc_ub2.so python scripts/lab/sweep.py $O/sweep_ub2.json 4096 $SEEDS "ub2:" 2>&1 | grep median | tee -a $O/sweep.log
