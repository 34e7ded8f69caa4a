#!/bin/bash
# GPU run r2a: sanity, baseline bench + profile, zero-code experiments (env knobs), per-task timing dumps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
b() { tag=$1; shift; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1]); print("$tag", round(d["value"]), round(d["ms_per_step"],3), d["tiers"])
except Exception as e: print("$tag FAILED", e)
PY
}
b base X=1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --episodes 16384 > $O/bench_base16k.json 2>$O/bench_base16k.err; tail -c 600 $O/bench_base16k.json
b w4096nw8 STMPC_TIERS=4096,8192 STMPC_NW=8,8 STMPC_PEN_CELLS=2048,4096
b w4096nw4 STMPC_TIERS=4096,8192 STMPC_NW=4,8 STMPC_PEN_CELLS=2048,4096
b cap64 STMPC_BAND_CAP=64
b cap128 STMPC_BAND_CAP=128
b cap200 STMPC_BAND_CAP=200
b nosplit STMPC_SPLIT=0
for cap in 300 128 64; do STMPC_BAND_CAP=$cap STMPC_LIB=$PWD/variants/libstmpc_times.so python scripts/lab/times_dump.py $O/times_cap$cap.bin; done
STMPC_TIERS=4096,8192 STMPC_NW=8,8 STMPC_PEN_CELLS=2048,4096 STMPC_LIB=$PWD/variants/libstmpc_times.so python scripts/lab/times_dump.py $O/times_w4096nw8.bin
# full default bench (with cpu baseline) + rocprof trace of the same command
python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 1500 $O/bench_full.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_base -o base -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_base.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof_base -name "*kernel_stats*" | head; for f in $(find $O/prof_base -name "*kernel_stats.csv"); do head -8 $f; done
