#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3q; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "golden or guided or oracle_seeded or overflow or config4 or bounded" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
SEEDS=1000,1,2,3,4,5,6,7
python scripts/lab/sweep.py $O/sweep.json 4096 $SEEDS "fan16:" 2>&1 | grep "median\|DIFFER" | tee $O/sweep.log
python scripts/lab/sweep.py $O/sweep8k.json 8192 1000,1,2 "fan16_8k:" 2>&1 | grep "median\|DIFFER" | tee -a $O/sweep.log
python scripts/lab/sweep.py $O/sweep16k.json 16384 1000 "fan16_16k:" 2>&1 | grep "median\|DIFFER" | tee -a $O/sweep.log
STMPC_LIB=$PWD/variants/libstmpc_times.so python scripts/lab/times_dump.py $O/times.bin > /dev/null 2>&1 || true
