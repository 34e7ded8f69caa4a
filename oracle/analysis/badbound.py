"""Episodes whose GPU pre-pass bound was far above the answer: which pre-pass design finds a usable bound?
Inputs: gpurun_out/times_ub.bin, times_ub_cost.npy, times_ub_bt.npy written on a GPU box by
  STMPC_LIB=variants/libstmpc_times.so python scripts/lab/ub_dump.py   (timing build: scripts/lab/mk_times.py).  Analysis infrastructure."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab
lab.build(); lab._init("h40a21")
t = np.fromfile('gpurun_out/times_ub.bin', dtype=np.uint64).reshape(-1, 16)
U = t[:, 13].copy().view(np.float64)
cost = np.load('gpurun_out/times_ub_cost.npy'); bt = np.load('gpurun_out/times_ub_bt.npy')
ok = (bt == 39) & np.isfinite(U) & (U > 0) & ((t[:, 14] & 1) > 0)
rel = np.where(ok, U / np.maximum(cost, 1e-9) - 1, 0)
bad = np.nonzero(rel > 0.2)[0]
print("bad-bound episodes:", bad.size, "of", ok.sum())
band = 1800.0
variants = {
    "hard": dict(band=band, cap=300, hs=1),
    "hard x4 cap600": dict(band=4 * band, cap=600, hs=1),
    "thin b7200 cap300": dict(band=7200.0, cap=300, hs=1, hmode=8),
    "thin b7200 cap600": dict(band=7200.0, cap=600, hs=1, hmode=8),
    "thin b20000 cap300": dict(band=20000.0, cap=300, hs=1, hmode=8),
    "thin b20000 cap600": dict(band=20000.0, cap=600, hs=1, hmode=8),
    "thin b1e9 cap600": dict(band=1e9, cap=600, hs=1, hmode=8),
    "soft x4": dict(band=4 * band, cap=300, hs=0),
}
sel = bad[:96]
res = {k: [] for k in variants}
nodes = {k: [] for k in variants}
for i in sel:
    g, v0, a0 = lab.grid_of(int(i))
    for k, kw in variants.items():
        o = lab.run_pass(g, v0, a0, **kw)
        res[k].append(o.cost / cost[i] - 1 if o.complete else np.inf)
        nodes[k].append(o.nodes)
for k in variants:
    r = np.array(res[k])
    print("%-18s complete %2d/%d  within 5%%: %2d  within 50%%: %2d   median nodes %d" % (k, np.isfinite(r).sum(), len(sel), (r < 0.05).sum(), (r < 0.5).sum(), np.median(nodes[k])))
print("GPU bound / cost - 1 for these:", np.round(rel[sel][:12], 2))
