#!/bin/bash
# A/B of builds of the library on the BASELINE workload: usage gpu_ab.sh <out dir under gpurun_out> <n> <seeds> <lib|-> ...   ("-" = the in-tree build)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; N=$2; SEEDS=$3; shift 3; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  tag=$(basename $lib .so); tag=${tag#libstmpc_}; [ "$lib" = "-" ] && tag=tree
  if [ "$lib" = "-" ]; then unset STMPC_LIB; else export STMPC_LIB=$lib; fi
  timeout 150 python scripts/lab/sweep.py $O/s_${tag}_$rep.json $N $SEEDS "$tag:" 2>&1 | grep -v amdgpu.ids >> $O/sweep.log
done; done
grep "median\|DIFFER" $O/sweep.log
python - <<PY
import json,glob,collections
d=collections.defaultdict(dict)
for f in sorted(glob.glob("$O/s_*_1.json")):
    for r in json.load(open(f)): d[r["seed"]][r["tag"]]=r["digest"]
bad=[s for s,v in d.items() if len(set(v.values()))>1]
print("digests equal across builds" if not bad else "DIGESTS DIFFER for seeds %s"%bad)
PY
