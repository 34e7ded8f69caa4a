#!/bin/bash
# Interleaved A/B of two builds of the library on one GPU box: <rounds> alternations of scripts/lab/sweep.py with variants/libstmpc_<base>.so
# and the product build, the same seeds; prints every seed-median and the two means.   usage: scripts/lab/ab.sh <tag> <base variant> [rounds] [n] [seeds]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
tag=$1; base=$2; rounds=${3:-4}; n=${4:-4096}; seeds=${5:-1000,1,2,3,4,5,6,7}
O=gpurun_out/$tag; mkdir -p $O
for r in $(seq 1 $rounds); do
  STMPC_LIB=$PWD/variants/libstmpc_$base.so python scripts/lab/sweep.py $O/a$r.json $n $seeds "base$r:" 2>&1 | grep "median\|DIFFER" | tee -a $O/ab.log
  python scripts/lab/sweep.py $O/b$r.json $n $seeds "new$r:" 2>&1 | grep "median\|DIFFER" | tee -a $O/ab.log
done
python - $O/ab.log <<'PY'
import re, sys
a, b = [], []
for l in open(sys.argv[1]):
    m = re.match(r"(base|new)\d+\s+seed-median of medians ([\d.]+) ms", l)
    if m: (a if m.group(1) == "base" else b).append(float(m.group(2)))
import statistics as st
print("base mean %.4f ms (%d runs)   new mean %.4f ms (%d runs)   new/base %.4f" % (st.mean(a), len(a), st.mean(b), len(b), st.mean(b) / st.mean(a)))
PY
