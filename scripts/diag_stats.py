#!/usr/bin/env python3
"""Print solver statistics (tier occupancy, expanded nodes, timings) for one batch. GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, st, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "default"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
if wl == "h40a21":
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
p = _capi.Params.from_settings(pkg.Settings)
ego, k, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=1000)
ctx = _capi.Context(0)
for rep in range(2):
    st.solve_arrays(ego, k, ox, ov, p, ctx)
s = ctx.stats()
print(wl, {k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in s.items()},
      "nodes/solve exact=%.0f bound=%.0f" % (s["nodes_exact"] / n, s["nodes_bound"] / n))
