#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3ai; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash scripts/gpu_ab.sh r3ai_ab 4096 1000,1,2,3,4,5,6,7 variants/libstmpc_base.so -
bash scripts/gpu_ab_wl.sh r3ai_def default variants/libstmpc_base.so -
