"""Does the growth of the layer minimum flag a poor pre-pass bound?  Same inputs as badbound.py.  Analysis infrastructure."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab
lab.build(); lab._init("h40a21")
cost = np.load('gpurun_out/times_ub_cost.npy'); bt = np.load('gpurun_out/times_ub_bt.npy')
rows = []
for i in range(1400):
    if bt[i] != 39: continue
    g, v0, a0 = lab.grid_of(i)
    o = lab.run_pass(g, v0, a0, band=1800.0, cap=300, hs=1)
    if not o.complete: continue
    km = np.array(o.lay_kmin[:39]); inc = np.diff(km)
    rel = o.cost / cost[i] - 1
    # features: largest increment after layer 4, relative to the median increment; final cost over layer-20 minimum
    rows.append((rel, inc[4:].max(), inc[4:].max() / max(np.median(inc[4:]), 1e-9), (o.cost - km[38]), o.cost / max(km[20], 1.0), inc[-5:].sum()))
r = np.array(rows); bad = r[:, 0] > 0.2
print("complete episodes", len(r), "bad", bad.sum())
for k, nm in enumerate(["max increment", "max/median increment", "final - last layer min", "final / min at layer 20", "sum of last 5 increments"], 1):
    g_, b_ = r[~bad, k], r[bad, k]
    print("%-26s good q50 %.1f q90 %.1f q99 %.1f | bad q10 %.1f q50 %.1f q90 %.1f" % (nm, *np.quantile(g_, [.5, .9, .99]), *np.quantile(b_, [.1, .5, .9])))
    for th in np.quantile(g_, [.9, .95, .98]):
        print("      threshold %.1f: flags %d good, %d of %d bad" % (th, (g_ > th).sum(), (b_ > th).sum(), bad.sum()))
