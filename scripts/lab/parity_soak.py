"""Soak: many further state seeds of both lattices, every episode against the oracle (beyond parity_wide.py / parity_narrow.py).
usage: parity_soak.py [first seed, default 200] [number of seeds, default 24] [n, default 4096]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, st, synth
from oracle import st_oracle as orc
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ctx = _capi.Context(0)
bad = tot = trunc = 0
for wl in ("h40a21", "default"):
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    if wl == "h40a21":
        pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    for seed in range(s0, s0 + ns):
        kk = (6, 6, 3, 12)[seed % 4]
        ego, k, ox, ov = synth.generate_states(n, k=kk, kmax=16, seed=seed, vary_k=(seed % 4 >= 2), dt=p.dt)
        res = st.solve_arrays(ego, k, ox, ov, p, ctx)
        ref = orc.solve_batch(op, ego, k, ox, ov, solver="layered", nthreads=16)
        ok = all(np.array_equal(res[key], ref[key]) for key in ("path_idx", "best_t", "cost", "crash"))
        bad += not ok; tot += n; trunc += int((ref["best_t"] < ref["path_idx"].shape[1] - 1).sum())
        if not ok:
            print("%s seed %d: DIFFERS" % (wl, seed), flush=True)
    print("%s: %d seeds x %d episodes done, batches that differ so far: %d" % (wl, ns, n, bad), flush=True)
print("soak: %d episodes (%d truncated paths), %d batches differ; library %s" % (tot, trunc, bad, _capi.backend_info()))
sys.exit(1 if bad else 0)
