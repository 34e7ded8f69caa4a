#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash profiles/run_profile.sh <tag> <bench args...>
# Writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<tag>/; summaries are then
# copied into profiles/ by hand (see profiles/README.md).
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python $REPO/bench.py "$@" --no-cpu-baseline > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $out/pmc1 -o pmc1 -- python $REPO/bench.py "$@" --no-cpu-baseline > $out/bench_pmc1.json 2> $out/pmc1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES -d $out/pmc2 -o pmc2 -- python $REPO/bench.py "$@" --no-cpu-baseline > $out/bench_pmc2.json 2> $out/pmc2.err
rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o pmc3 -- python $REPO/bench.py "$@" --no-cpu-baseline > $out/bench_pmc3.json 2> $out/pmc3.err
rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o pmc4 -- python $REPO/bench.py "$@" --no-cpu-baseline > $out/bench_pmc4.json 2> $out/pmc4.err
cd $REPO
find $out -name "*.csv" | head -40
