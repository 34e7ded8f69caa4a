#!/usr/bin/env python3
"""Run do_st_control for one batch (lattice search + QP re-sampling); use under rocprofv3 --kernel-trace. GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "default"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
if wl == "h40a21":
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
p = _capi.Params.from_settings(pkg.Settings)
ego, k, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=1000)
ctx = _capi.Context(0)
for rep in range(3):
    r = ctx.st_control_batch(p, pkg.Settings.TICK_LENGTH, ego, k, ox, ov, want_paths=True)
print(wl, n, "fine_len", np.bincount(r["fine_len"].clip(0)).nonzero()[0].tolist(), "speed mean", r["speed"].mean(), ctx.stats())
