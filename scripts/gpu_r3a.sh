#!/bin/bash
# GPU run r3a: parity of the single-precision bounding pass, A/B against the round-2 build, node-cap sweep, reserved-CU sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
SEEDS=1000,1,2,3,4
STMPC_LIB=$PWD/variants/libstmpc_r2.so timeout 300 python scripts/lab/sweep.py $O/sweep_r2.json 4096 $SEEDS "r2:" 2>&1 | tee $O/sweep_r2.log
timeout 600 python scripts/lab/sweep.py $O/sweep_new.json 4096 $SEEDS "new:" "cap200:STMPC_BAND_CAP=200" "cap450:STMPC_BAND_CAP=450" "cap600:STMPC_BAND_CAP=600" "cap900:STMPC_BAND_CAP=900" "cap1200:STMPC_BAND_CAP=1200" \
   "cap600b16:STMPC_BAND_CAP=600,STMPC_BAND=3600" "cap900b16:STMPC_BAND_CAP=900,STMPC_BAND=3600" "nores:STMPC_RESUME=0" "noovl:STMPC_OVERLAP=0" 2>&1 | tee $O/sweep_new.log
timeout 300 python scripts/lab/sweep.py $O/sweep_new8k.json 8192 1000,1,2 "new8k:" "cap600_8k:STMPC_BAND_CAP=600" "cap900_8k:STMPC_BAND_CAP=900" 2>&1 | tee $O/sweep_new8k.log
STMPC_LIB=$PWD/variants/libstmpc_r2.so timeout 300 python scripts/lab/sweep.py $O/sweep_r2_8k.json 8192 1000,1,2 "r2_8k:" 2>&1 | tee $O/sweep_r2_8k.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; tail -c 700 $O/bench_new.json
# reserved compute units (CU-masked streams); guarded by a timeout
timeout 240 python scripts/lab/sweep.py $O/sweep_cu.json 4096 $SEEDS "res8:STMPC_CU_RESERVE=8" "res16:STMPC_CU_RESERVE=16" "res32:STMPC_CU_RESERVE=32" "res16c600:STMPC_CU_RESERVE=16,STMPC_BAND_CAP=600" "res32c600:STMPC_CU_RESERVE=32,STMPC_BAND_CAP=600" 2>&1 | tee $O/sweep_cu.log
echo done
