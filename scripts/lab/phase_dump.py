"""Phase profile of one batch with an analysis build (STMPC_LIB=variants/libstmpc_phase.so).  usage: phase_dump.py <out.txt> [n]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, st, synth
pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
if os.environ.get("PHASE_WORKLOAD", "h40a21") == "h40a21":
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
p = _capi.Params.from_settings(pkg.Settings)
out = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ego, k, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=1000)
ctx = _capi.Context(0)
st.solve_arrays(ego, k, ox, ov, p, ctx)
os.environ["STMPC_PHASE_DUMP"] = out
st.solve_arrays(ego, k, ox, ov, p, ctx)
s = ctx.stats()
a = np.loadtxt(out)
names = ["setup", "layer-setup", "scan+S1", "list/best", "src+range+filter", "reduce+B1", "geometry", "init+B2", "stageA-rest+B3", "stageB+B4", "stageC", "layer-end", "#rounds", "#batches"]
print(out, "solve_ms %.3f nodes exact %d bound %d" % (s["solve_ms"], s["nodes_exact"], s["nodes_bound"]))
for m, nm in enumerate(("EXACT", "BOUND")):
    tot = a[m][:12].sum()
    print("    rounds %d, candidate batches %d, candidates offered %d of %d lane-slots" % (a[m][12], a[m][13], a[m][14], a[m][15]))
    print("  %s total %.1f Mcycles (thread 0 of all workgroups)" % (nm, tot / 1e6))
    for k_, pn in enumerate(names[:12]):
        print("    %-18s %6.2f %%" % (pn, 100 * a[m][k_] / max(tot, 1)))
    if a.shape[0] >= 4:
        w = a[2 + m]
        print("    waiting at barriers (lane 0 of every wave, share of the waves' time): " + "  ".join("%s %.1f %%" % (nm_, 100 * w[k_] / max(w[15], 1)) for k_, nm_ in enumerate(("S1", "B1", "B2", "B3", "B4", "S2"))) + "  all %.1f %%" % (100 * w[:6].sum() / max(w[15], 1)))
