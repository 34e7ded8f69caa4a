#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3m; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
STMPC_BP16=1 python -m pytest tests -m gpu -x -q -k "golden or overflow or oracle_seeded or config4" > $O/pytest_gpu_bp16.log 2>&1; echo "pytest(bp16) rc=$?"; tail -2 $O/pytest_gpu_bp16.log
SEEDS=1000,1,2,3,4,5,6,7
timeout 900 python scripts/lab/sweep.py $O/sweep.json 4096 $SEEDS "rel8:" "bp16:STMPC_BP16=1" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --seeds= > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; for f in $(find $O/trace -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; done
cd /tmp && rocprofv3 --pmc WRITE_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_w -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0 --seeds= > $GRAFT_REPO_ROOT/$O/pmc_w.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<PY
import csv, glob, collections
tot=collections.defaultdict(float)
for f in glob.glob("$O/pmc_w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)): tot[r["Kernel_Name"].split("(")[0][:60]] += float(r["Counter_Value"])
for k,v in tot.items(): print("WRITE_SIZE KiB/step %-62s %.0f" % (k, v/13))
PY
