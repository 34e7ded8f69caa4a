#!/bin/bash
# Profile of the default bench command on the GPU box: kernel trace + counter passes (each in its own run) + the
# analysis build's phase/candidate counters.  usage: scripts/profile_gpu.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/
tag=${1:-r2}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/prof_$tag; mkdir -p $O
export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --seeds= $*"
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass -f csv -d $O/pmc_$n -o p -- $CMD > $O/pmc_$n.log 2>&1
done
cd $R
WL=h40a21; case "$*" in *default*|*control*) WL=default;; esac
if [ -f variants/libstmpc_phase.so ]; then PHASE_WORKLOAD=$WL STMPC_LIB=$R/variants/libstmpc_phase.so python scripts/lab/phase_dump.py $O/phase.txt > $O/phase_report.txt 2>&1; fi
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --seeds= $* > $O/bench_line.json 2>/dev/null
find $O -name "*.csv" | head -30
