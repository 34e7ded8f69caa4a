#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7,8,9
specs=("base:")
for cap in 300 450 600; do for r in "1.002,1.02,1.3" "1.004,1.03,1.3" "1.008,1.05,1.3"; do specs+=("c${cap}r${r%%,*}:STMPC_BAND_CAP=$cap;STMPC_RETRY=$r"); done; done
specs+=("c450:STMPC_BAND_CAP=450" "c600:STMPC_BAND_CAP=600" "c375:STMPC_BAND_CAP=375" "c525:STMPC_BAND_CAP=525")
timeout 900 python scripts/lab/sweep.py $O/sweep.json 4096 $SEEDS "${specs[@]}" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
