#!/bin/bash
# Regenerates profiles/<round>/ from the current sources (run in the build container; two gpurun calls, ~5 GPU-minutes).
# Needs the library and the two analysis builds of the SAME sources:  python -c "import __graft_entry__ as g; g.build()";
# scripts/lab/mk_variant.sh phase -DSTMPC_PHASE_PROF;  python scripts/lab/mk_times.py;  bash scripts/isa/spill_report.sh profiles/<round>
# usage: scripts/evidence.sh r5
set -e
round=${1:-r5}; P=profiles/$round; R=gpurun_out/ev1; G=/usr/local/graft/bin/gpurun
cd "$(dirname "$0")/.."
H=$(python -c 'import rl_mpc_lanemerging_amd as p; print(p.build.source_hash())')
rm -rf gpurun_out/ev1 gpurun_out/ev2 gpurun_out/prof_$round gpurun_out/prof_${round}d
# pass 1: suite, smoke, the workloads without counters, profiles (kernel trace + --pmc passes + phase build), parity sweeps, episode tables, schedule
$G --timeout 2400 -- "bash scripts/gpu_run.sh ev1 env=STMPC_TEST_ARTEFACTS=gpurun_out/ev1 tests smoke bench=--workload+control bench=--workload+combined bench=--workload+episodes prof=$round prof=${round}d+--workload+default py=scripts/lab/parity_wide.py py=scripts/lab/parity_narrow.py py=scripts/lab/combined_episodes.py+1024+gpurun_out/ev1/combined_episodes.json py=scripts/lab/crash_probe.py+2048 env=STMPC_LIB=/root/repo/variants/libstmpc_times.so py=scripts/lab/times_dump.py+gpurun_out/ev1/t.bin py=scripts/lab/times_tail.py+gpurun_out/ev1/t.bin" 2>&1 | grep "^\[" | cut -c1-220
# stage numbers of pass 1: 1 env, 2 tests, 3 smoke, 4-6 bench, 7-8 prof, 9-12 py, 13 env, 14-15 py (timing build)
cp $R/bench_4.json $P/bench_control.json; cp $R/bench_5.json $P/bench_combined.json; cp $R/bench_6.json $P/bench_episodes.json
cp $R/config4_full_size.json $R/combined_episodes.json $P/
{ echo "# row f3, pure ST controller, 2048 episodes per traffic density against the reference's rows (scripts/lab/crash_probe.py); source $H"; grep -v amdgpu.ids $R/py_12.log; } > $P/st_episodes.txt
grep -v amdgpu.ids $R/py_9.log > $P/parity_wide.txt; grep -v amdgpu.ids $R/py_10.log > $P/parity_narrow.txt
{ echo "# per-task schedule of one N=4096 step of the timing-instrumented build (scripts/lab/mk_times.py + times_dump.py + times_tail.py), microseconds from the first task; source $H"; grep -v amdgpu.ids $R/py_15.log; } > $P/schedule.txt
python scripts/profile_summarize.py $round $P h40a21 > /dev/null; python scripts/profile_summarize.py ${round}d $P default > /dev/null; mv $P/${round}d_summary.txt $P/${round}_default_summary.txt
# pass 2: the solver's bench lines, now against counters of this build (measured.json)
$G --timeout 1200 -- 'bash scripts/gpu_run.sh ev2 bench bench=--workload+default bench=--episodes+8192 bench=--episodes+16384' 2>&1 | grep "^\[" | cut -c1-220
cp gpurun_out/ev2/bench_1.json $P/bench_h40a21.json; cp gpurun_out/ev2/bench_2.json $P/bench_default.json; cp gpurun_out/ev2/bench_3.json $P/bench_h40a21_n8k.json; cp gpurun_out/ev2/bench_4.json $P/bench_h40a21_n16k.json
echo "files of $P without the source hash $H:"; grep -L "$H" $P/bench_*.json $P/config4_full_size.json $P/measured.json $P/schedule.txt $P/spill_placement.txt || true
