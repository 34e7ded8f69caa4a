#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (scripts/profile_gpu.sh) into profiles/<round>/<tag>_summary.txt and the
per-workload entry of profiles/<round>/measured.json that bench.py reads (HBM bytes per step, candidates per node).
usage: scripts/profile_summarize.py <tag> <round dir> <workload>"""
import csv, glob, json, os, sys
from collections import defaultdict
tag, rdir, workload = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs(rdir, exist_ok=True)
lines = ["# rocprofv3 summary of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipelined 0` (%s), scripts/profile_gpu.sh" % workload]
steps = 13      # 2 warm-up + 10 timed + 1 statistics step
# kernel trace
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
dominant = None
if ks:
    lines.append("# --kernel-trace --stats (ns)")
    lines += [l.rstrip() for l in open(ks[0])][:12]
    # the dominant kernel = the lattice-DP launch every step makes exactly once (the first window; a side launch of the second window runs
    # alongside it and its duration is mostly waiting): the k_solve row with one call per step and the largest average
    rows = [r for r in csv.DictReader(open(ks[0])) if "k_solve" in r["Name"] and int(r["Calls"]) == steps]
    if rows:
        r = max(rows, key=lambda r_: float(r_["AverageNs"]))
        dominant = {"dominant_kernel": r["Name"].split("(")[0].replace("void ", ""), "dominant_kernel_calls": int(r["Calls"]),
                    "dominant_kernel_avg_ms": float(r["AverageNs"]) / 1e6, "dominant_kernel_min_ms": float(r["MinNs"]) / 1e6, "dominant_kernel_max_ms": float(r["MaxNs"]) / 1e6}
def pmc(n):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_" + n, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out
tot = defaultdict(float)
for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT"):
    d = pmc(n)
    if not d: continue
    lines.append("# --pmc pass %s (kernel, counter, dispatches, sum over dispatches / %d steps)" % (n, steps))
    for k in sorted(d):
        for c in sorted(d[k]):
            v = d[k][c]
            lines.append("%s,%s,%d,%.1f" % (k, c, len(v), sum(v) / steps))
            tot[c] += sum(v) / steps
meas = {}
if dominant:
    meas.update(dominant)
    if workload in ("h40a21", "default"):
        # BASELINE's HBM roofline from the trace: algorithmic bytes per launch (140 B state + 16 B action / cost / layer + 4 H of path per solve,
        # 4096 solves) / the dominant kernel's average duration / 8 TB/s
        Hh = 40 if workload == "h40a21" else 18
        meas["algorithmic_bytes_per_launch"] = (156 + 4 * Hh) * 4096
        meas["hbm_roofline_achieved_gbs"] = meas["algorithmic_bytes_per_launch"] / (dominant["dominant_kernel_avg_ms"] * 1e-3) / 1e9
        meas["hbm_roofline_frac"] = meas["hbm_roofline_achieved_gbs"] / 8000.0
if "FETCH_SIZE" in tot or "WRITE_SIZE" in tot:
    # FETCH_SIZE / WRITE_SIZE are in KiB?  rocprofv3 reports them in kilobytes (derived: TCC_EA0_RDREQ*64/1024 ...); gfx950 correction x2 on FETCH (MI355X_MICROARCH.md, HBM section)
    fetch_b, write_b = tot.get("FETCH_SIZE", 0.0) * 1024.0, tot.get("WRITE_SIZE", 0.0) * 1024.0
    meas["hbm_bytes_per_step"] = 2.0 * fetch_b + write_b
    meas["hbm_source"] = ("profiles/%s: FETCH_SIZE %.0f KiB x2 (gfx950 correction) + WRITE_SIZE %.0f KiB per step, separate --pmc passes, "
                          "summed over all kernels of one step" % (os.path.basename(rdir), tot.get("FETCH_SIZE", 0), tot.get("WRITE_SIZE", 0)))
    lines.append("# HBM traffic per step: FETCH %.1f MB (x2 corrected: %.1f MB) + WRITE %.1f MB" % (fetch_b / 1e6, 2 * fetch_b / 1e6, write_b / 1e6))
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    if c in tot: meas["per_step_" + c] = tot[c]
ph = os.path.join(src, "phase.txt")
if os.path.exists(ph):
    import numpy as np
    a = np.loadtxt(ph)
    rep = open(os.path.join(src, "phase_report.txt")).read()
    lines.append("# analysis build (-DSTMPC_PHASE_PROF), one batch: phase shares of thread 0's clock; candidate counters")
    lines += ["# " + l for l in rep.splitlines()]
    import re
    m = re.search(r"nodes exact (\d+) bound (\d+)", rep)
    if m:
        nodes = int(m.group(1)) + int(m.group(2))
        cand = a[0][14] + a[1][14]; slots = a[0][15] + a[1][15]
        meas["candidates_per_node"] = cand / nodes
        meas["lane_slots_per_candidate"] = slots / max(cand, 1)
        lines.append("# candidates offered per expanded node: %.2f ; executed lane-slots per offered candidate: %.2f" % (cand / nodes, slots / max(cand, 1)))
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl):
    try:
        d = json.loads(open(bl).read().strip().splitlines()[-1]); meas["bench_line_value"] = d["value"]; meas["bench_line_ms_per_step"] = d["ms_per_step"]; meas["bench_line_kernel_ms"] = d["roofline"]["kernel_ms"]; lines.append("# same-run bench line: %.0f %s, %.3f ms/step, kernel_ms %.3f" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
    except Exception: pass
open(os.path.join(rdir, tag + "_summary.txt"), "w").write("\n".join(lines) + "\n")
mj = os.path.join(rdir, "measured.json")
allm = json.load(open(mj)) if os.path.exists(mj) else {}
# the sources the profiled library was built from (the same-run bench line carries the library's own hash): bench.py marks these
# counters stale when the running library differs
try:
    meas["csrc_hash"] = json.loads(open(bl).read().strip().splitlines()[-1])["library"]["csrc_hash"]
except Exception:
    sys.path.insert(0, os.getcwd())
    import rl_mpc_lanemerging_amd as _pkg
    meas["csrc_hash"] = _pkg.build.source_hash()
allm[workload] = meas
json.dump(allm, open(mj, "w"), indent=1, sort_keys=True)
print("\n".join(lines[:40])); print(json.dumps(meas, indent=1))
