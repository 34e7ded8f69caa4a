#!/bin/bash
# dense ordinary bounding attempts (band_pass): parity, then on/off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3ag; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q -k "golden or guided or oracle_seeded or overflow or config4 or bounded or other_parameter" > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest.log
SEEDS=1000,1,2,3,4,5,6,7
timeout 300 python scripts/lab/sweep.py $O/s.json 4096 $SEEDS "dense:" "nodense:STMPC_BAND_DENSE=0" 2>&1 < /dev/null | grep -v amdgpu.ids > $O/sweep.log
grep "median\|DIFFER" $O/sweep.log; grep "seed  1000" $O/sweep.log
timeout 100 python scripts/lab/sweep.py $O/s8k.json 8192 1000,1,2 "dense8k:" "nodense8k:STMPC_BAND_DENSE=0" 2>&1 < /dev/null | grep "median\|DIFFER" | tee -a $O/sweep.log
