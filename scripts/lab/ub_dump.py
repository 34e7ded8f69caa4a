"""One H40/A21 batch with the timing build: per-episode stamps + the pre-pass bound (slot 13) -> gpurun_out/times_ub.bin, and the
solver's cost / best_t -> gpurun_out/times_ub_cost.npy, times_ub_bt.npy (inputs of oracle/analysis/badbound.py)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, st, synth
pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
p = _capi.Params.from_settings(pkg.Settings)
n=4096
ego, k, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=1000)
ctx = _capi.Context(0)
st.solve_arrays(ego, k, ox, ov, p, ctx)
os.environ["STMPC_DUMP_TIMES"] = "gpurun_out/times_ub.bin"
r = st.solve_arrays(ego, k, ox, ov, p, ctx)
np.save("gpurun_out/times_ub_cost.npy", r["cost"]); np.save("gpurun_out/times_ub_bt.npy", r["best_t"])
