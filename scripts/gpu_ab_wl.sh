#!/bin/bash
# A/B of builds on a bench workload: usage gpu_ab_wl.sh <out dir> <workload> <lib|-> ...
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; WL=$2; shift 2; mkdir -p $O
for rep in 1 2 3; do
for lib in "$@"; do
  tag=$(basename $lib .so); tag=${tag#libstmpc_}; [ "$lib" = "-" ] && tag=tree
  if [ "$lib" = "-" ]; then unset STMPC_LIB; else export STMPC_LIB=$lib; fi
  timeout 120 python bench.py --workload $WL --no-cpu-baseline --seeds= --steps 40 --warmup 5 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', '$WL', round(d['value']), d['unit'], round(d['ms_per_step'],4), d.get('parity_vs_oracle',{}).get('path_idx'))"
done; done
