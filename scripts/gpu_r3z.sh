#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3z; mkdir -p $O
timeout 200 python -m pytest tests -m gpu -x -q -k "golden or guided or oracle_seeded or overflow or config4 or bounded or other_parameter" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
SEEDS=1000,1,2,3,4,5,6,7,8,9,10,11
timeout 200 python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "bsplit:" "nobsplit:STMPC_BSPLIT=0" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | grep "median\|DIFFER"
timeout 100 python scripts/lab/sweep.py $O/s8k.json 8192 1000,1,2 "bsplit8k:" "nobsplit8k:STMPC_BSPLIT=0" 2>&1 | grep "median\|DIFFER" | tee -a $O/sweep.log
timeout 100 python scripts/lab/sweep.py $O/s16k.json 16384 1000 "bsplit16k:" "nobsplit16k:STMPC_BSPLIT=0" 2>&1 | grep "median\|DIFFER" | tee -a $O/sweep.log
