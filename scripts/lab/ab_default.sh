#!/bin/bash
# Interleaved A/B of a variant build against the product build on the reference's own lattice (bench.py --workload default / combined).
# usage: scripts/lab/ab_default.sh <variant> [rounds]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
v=$1; rounds=${2:-3}
one() { python bench.py --workload $1 --no-cpu-baseline --pipelined 0 --seeds= --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
for r in $(seq 1 $rounds); do
  echo "round $r  default: product $(one default)  variant $(STMPC_LIB=$PWD/variants/libstmpc_$v.so one default)   combined: product $(one combined)  variant $(STMPC_LIB=$PWD/variants/libstmpc_$v.so one combined)"
done
