#!/bin/bash
# Interleaved comparison of several builds of the library on one GPU box: <rounds> alternations of scripts/lab/sweep.py over variants/libstmpc_<name>.so
# ('-' = the product build), the same seeds; prints every seed-median and the mean per build relative to the first.
# usage: scripts/lab/abn.sh <tag> <rounds> <n> <seeds> <name> [<name> ...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
tag=$1; rounds=$2; n=$3; seeds=$4; shift 4
O=gpurun_out/$tag; mkdir -p $O; : > $O/ab.log
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset STMPC_LIB; name=product; else export STMPC_LIB=$PWD/variants/libstmpc_$v.so; name=$v; fi
    python scripts/lab/sweep.py $O/${name}_$r.json $n $seeds "$name:" 2>&1 | grep "median\|DIFFER" | tee -a $O/ab.log
  done
done
python - $O/ab.log "$@" <<'PY'
import re, sys, statistics as st
rows = {}
for l in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+seed-median of medians ([\d.]+) ms", l)
    if m: rows.setdefault(m.group(1), []).append(float(m.group(2)))
names = [("product" if v == "-" else v) for v in sys.argv[2:]]
base = st.mean(rows[names[0]])
for nm in names:
    print("%-12s mean %.4f ms (%d runs)  / %s = %.4f" % (nm, st.mean(rows[nm]), len(rows[nm]), names[0], st.mean(rows[nm]) / base))
import json, glob, os
# digests: every build must return the same bits
d = {}
for f in glob.glob(os.path.join(os.path.dirname(sys.argv[1]), "*_1.json")):
    for r in json.load(open(f)): d.setdefault(r["seed"], set()).add(r["digest"])
print("results identical across builds:", all(len(v) == 1 for v in d.values()))
PY
