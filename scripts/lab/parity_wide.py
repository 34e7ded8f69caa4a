"""Beyond the suite: every episode of several batches against the oracle on the GPU box -- further state seeds of the benchmark lattice, varying
vehicle counts (up to 3 / 12 / 20, a tenth with a blocked start), and the reference's own lattice.  usage: parity_wide.py [n]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, st, synth
from oracle import st_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bad = 0
ctx = _capi.Context(0)
def check(tag, p, ego, k, ox, ov):
    global bad
    res = st.solve_arrays(ego, k, ox, ov, p, ctx)
    ref = orc.solve_batch(orc.OrcParams.from_dict(p.as_dict()), ego, k, ox, ov, solver="layered", nthreads=16)
    ok = all(np.array_equal(res[key], ref[key]) for key in ("path_idx", "best_t", "cost", "crash"))
    bad += not ok
    print("%-34s %5d episodes identical to the oracle: %s  (truncated paths %d, crash verdicts %d)" % (tag, len(ego), ok, int((ref["best_t"] < ref["path_idx"].shape[1] - 1).sum()), int(ref["crash"].sum())), flush=True)
pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
p = _capi.Params.from_settings(pkg.Settings)
for seed in (11, 12, 13, 14, 15, 16):
    check("h40a21 seed %d" % seed, p, *synth.generate_states(n, k=6, kmax=8, seed=seed))
for kk, kmax in ((3, 4), (12, 16), (20, 32)):
    check("h40a21 up to %d vehicles" % kk, p, *synth.generate_states(3000, k=kk, kmax=kmax, seed=100 + kk, vary_k=True, dt=p.dt))
pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
p = _capi.Params.from_settings(pkg.Settings)
for seed in (21, 22):
    check("reference lattice seed %d" % seed, p, *synth.generate_states(n, k=6, kmax=8, seed=seed))
check("reference lattice up to 20 vehicles", p, *synth.generate_states(3000, k=20, kmax=32, seed=123, vary_k=True, dt=p.dt))
sys.exit(1 if bad else 0)
