#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3o; mkdir -p $O
SEEDS=1000,1,2,3,4,5,6,7
python scripts/lab/sweep.py $O/sweep_base.json 4096 $SEEDS "base:" 2>&1 | grep -v amdgpu.ids | tee $O/sweep_base.log
STMPC_LIB=$PWD/variants/libstmpc_ilp88.so python scripts/lab/sweep.py $O/sweep_ilp.json 4096 $SEEDS "ilp88:" "ilp88_nw12:STMPC_NW=4,12" "ilp88_nw16:STMPC_NW=4,16" 2>&1 | grep -v amdgpu.ids | tee $O/sweep_ilp.log
python - <<PY
import json
a={(r["seed"]):r["digest"] for r in json.load(open("$O/sweep_base.json"))}
b=json.load(open("$O/sweep_ilp.json"))
print("digests equal across builds:", all(a[r["seed"]]==r["digest"] for r in b))
PY
