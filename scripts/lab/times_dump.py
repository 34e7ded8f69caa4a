"""Run one batch with the timing-instrumented build (STMPC_LIB=variants/libstmpc_times.so) and dump per-task stamps.
usage: STMPC_LIB=... python scripts/lab/times_dump.py <out.bin> [n] [seed]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, st, synth
pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
p = _capi.Params.from_settings(pkg.Settings)
out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
ego, k, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=seed)
ctx = _capi.Context(0)
st.solve_arrays(ego, k, ox, ov, p, ctx)
os.environ["STMPC_DUMP_TIMES"] = out
r = st.solve_arrays(ego, k, ox, ov, p, ctx)
s = ctx.stats()
np.save(out + ".cost.npy", r["cost"]); np.save(out + ".bt.npy", r["best_t"])
print(out, {q: s[q] for q in ("solve_ms", "fallback", "retries", "nodes_exact", "nodes_bound")})
