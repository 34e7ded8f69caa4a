#!/bin/bash
# One parameterised driver for everything that runs on the GPU box (replaces the one-off gpu_r*.sh scripts of earlier rounds).
# usage (through gpurun):  scripts/gpu_run.sh <tag> <stage> [<stage> ...]        -> gpurun_out/<tag>/
# stages:
#   tests[=<pytest -k expression>]            pytest -m gpu (all, or the selected tests)
#   smoke                                     __graft_entry__.smoke()
#   bench[=<bench.py args, '+' for spaces>]   one bench.py line -> bench_<i>.json
#   sweep=<n>/<seeds>/<lib|->/<spec>[/<spec>...]   scripts/lab/sweep.py with STMPC_LIB=variants/libstmpc_<lib>.so ('-' = the product build);
#                                             a spec is tag:ENV=V;ENV=V  (use ',' inside seeds, '+' for spaces)
#   trace[=<bench args>]                      rocprofv3 --kernel-trace --stats of a short bench run -> trace_<i>/
#   prof=<tag2>[+bench args]                  scripts/profile_gpu.sh <tag2> (kernel trace + 4 --pmc passes + phase build if present)
#   env=NAME=VALUE                            export a variable for the following stages (e.g. env=STMPC_LIB=/root/repo/variants/libstmpc_x.so)
#   py=<script.py>[+args]                     python <script> args   (stdout -> py_<i>.log)
# Every stage runs under its own `timeout`; a failing stage does not stop the following ones.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
i=0
for stage in "$@"; do
  i=$((i+1)); kind=${stage%%=*}; arg=""; [ "$kind" != "$stage" ] && arg=${stage#*=}
  arg_sp=${arg//+/ }
  t0=$(date +%s.%N)
  case $kind in
    tests)
      if [ -n "$arg" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$arg_sp" > $O/pytest_$i.log 2>&1
      else timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_$i.log 2>&1; fi
      echo "[$stage] rc=$? $(tail -1 $O/pytest_$i.log)";;
    smoke)
      timeout 300 python __graft_entry__.py smoke > $O/smoke_$i.log 2>&1; echo "[$stage] rc=$? $(tail -1 $O/smoke_$i.log)";;
    bench)
      timeout 900 python bench.py $arg_sp > $O/bench_$i.json 2> $O/bench_$i.err; rc=$?
      echo "[$stage] rc=$rc $(python - $O/bench_$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value %.0f %s  ms/step %.3f  seed-median %s  device_ms %s kernel_ms %s  parity %s  tiers %s" % (d["value"], d["unit"], d["ms_per_step"], d.get("value_seed_median"),
          d.get("device_ms_per_step"), d.get("roofline",{}).get("kernel_ms"), d.get("parity_vs_oracle"), d.get("tiers")))
except Exception as e: print("no line:", e)
PY
)";;
    sweep)
      IFS='/' read -r n seeds lib rest <<< "$arg"
      specs=(); IFS='/' read -ra parts <<< "$rest"; for s in "${parts[@]}"; do specs+=("${s//+/ }"); done
      [ ${#specs[@]} -eq 0 ] && specs=("base:")
      if [ "$lib" != "-" ]; then export STMPC_LIB=$PWD/variants/libstmpc_$lib.so; else unset STMPC_LIB; fi
      timeout 900 python scripts/lab/sweep.py $O/sweep_$i.json $n $seeds "${specs[@]}" 2>&1 | grep -v amdgpu.ids > $O/sweep_$i.log
      unset STMPC_LIB
      echo "[$stage]"; grep "median\|DIFFER" $O/sweep_$i.log;;
    trace)
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$O/trace_$i -o t -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --seeds= $arg_sp > $OLDPWD/$O/trace_$i.log 2>&1 )
      f=$(find $O/trace_$i -name "*kernel_stats.csv" | head -1)
      echo "[$stage]"; [ -n "$f" ] && head -8 "$f" | cut -c1-220;;
    prof)
      timeout 1500 bash scripts/profile_gpu.sh $arg_sp > $O/prof_$i.log 2>&1; echo "[$stage] rc=$?";;
    env)
      export "$arg_sp"; echo "[env] $arg_sp";;
    py)
      timeout 900 python $arg_sp > $O/py_$i.log 2>&1; echo "[$stage] rc=$? $(tail -2 $O/py_$i.log)";;
    *) echo "unknown stage $stage";;
  esac
  echo "   ($(python -c "print('%.1f' % ($(date +%s.%N) - $t0))") s)"
done
