#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3ae; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q -k "golden or guided or oracle_seeded or overflow or config4 or bounded or other_parameter" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash scripts/gpu_ab.sh r3ae_ab 4096 1000,1,2,3,4,5 variants/libstmpc_base.so -
STMPC_TUBE_KERNEL=1 timeout 100 python scripts/lab/sweep.py $O/tk.json 4096 1000,1,2 "tubek:" 2>&1 | grep "median\|DIFFER"
