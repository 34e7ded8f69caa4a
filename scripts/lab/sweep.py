"""In-process sweep of environment knobs on the H40/A21 benchmark batch (one import, one context per spec).
usage: [STMPC_LIB=...] python scripts/lab/sweep.py <out.json> <n> <seeds: 1000,1,2> "tag:ENV=V;ENV=V" ...
Per spec and seed: solve time (library events, min / median of 5), window overflows, retries, nodes; the result arrays of every spec
must equal those of the first spec (every knob is exact), and their digests go to <out.json> so that runs of different builds can
be compared as well."""
import hashlib, json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, st, synth

out_path, n = sys.argv[1], int(sys.argv[2])
seeds = [int(x) for x in sys.argv[3].split(",")]
specs = sys.argv[4:] or ["base:"]
pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
p = _capi.Params.from_settings(pkg.Settings)
states = {sd: synth.generate_states(n, k=6, kmax=8, seed=sd) for sd in seeds}
ref = {}
rows = []
for spec in specs:
    tag, _, envs = spec.partition(":")
    kv = [e.split("=", 1) for e in envs.replace(";", " ").split() if e]
    for k_, v_ in kv: os.environ[k_] = v_
    ctx = _capi.Context(0)
    meds = []
    for sd in seeds:
        ego, k, ox, ov = states[sd]
        r = st.solve_arrays(ego, k, ox, ov, p, ctx)
        ms = []
        for _ in range(5):
            r = st.solve_arrays(ego, k, ox, ov, p, ctx); ms.append(ctx.stats()["solve_ms"])
        s = ctx.stats()
        dig = hashlib.sha256(r["path_idx"].tobytes() + r["best_t"].tobytes() + r["cost"].tobytes() + r["crash"].tobytes()).hexdigest()[:16]
        same = True
        if sd in ref: same = dig == ref[sd]
        else: ref[sd] = dig
        meds.append(sorted(ms)[2])
        rows.append({"tag": tag, "seed": sd, "min_ms": min(ms), "med_ms": sorted(ms)[2], "fallback": s["fallback"], "retries": s["retries"],
                     "nodes_exact": s["nodes_exact"] / n, "nodes_bound": s["nodes_bound"] / n, "digest": dig, "same_as_first": same})
        print("%-14s seed %5d  min %.3f med %.3f  overflow %4d (last tier %4d) retries %4d guided %4d pairs %3d  nodes %.0f + %.0f  %s" % (tag, sd, min(ms), sorted(ms)[2], s["fallback"], s["hbm_tier"], s["retries"],
              s.get("guided", 0), s.get("pairs", 0), s["nodes_bound"] / n, s["nodes_exact"] / n, "" if same else "RESULTS DIFFER"), flush=True)
    print("%-14s seed-median of medians %.3f ms  (%.0f solves/s)" % (tag, float(np.median(meds)), n / float(np.median(meds)) * 1e3), flush=True)
    del ctx
    for k_, _ in kv: os.environ.pop(k_, None)
json.dump(rows, open(out_path, "w"), indent=0)
