#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
python scripts/lab/run_episodes.py 512 2>&1 | grep -v amdgpu.ids | tee $O/episodes.log
python -m pytest tests -m gpu -x -q --deselect tests/test_episodes.py > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
