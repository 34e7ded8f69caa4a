#!/bin/bash
# GPU run r3b: schedule (per-task stamps) and phase profile of the new bounding pass
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
for cap in 300 600; do STMPC_BAND_CAP=$cap STMPC_LIB=$PWD/variants/libstmpc_times.so python scripts/lab/times_dump.py $O/times_cap$cap.bin; done
python scripts/lab/times_ana.py $O/times_cap300.bin $O/times_cap600.bin | tee $O/times_ana.txt
STMPC_LIB=$PWD/variants/libstmpc_phase.so python scripts/lab/phase_dump.py $O/phase300.txt | tee $O/phase300_report.txt
STMPC_BAND_CAP=600 STMPC_LIB=$PWD/variants/libstmpc_phase.so python scripts/lab/phase_dump.py $O/phase600.txt | tee $O/phase600_report.txt
for w in 4 8 16; do STMPC_WAVES_PER_CU=$w python scripts/lab/sweep.py $O/sweep_w$w.json 16384 1000 "wpc$w:" 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep_wpc.log; done
