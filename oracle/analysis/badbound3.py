"""Aggregate quality of pre-pass designs over the first 1000 benchmark states: bounds more than 20 % / 2 % above the answer, bounds below it
(each costs a repeated exact pass), pre-pass nodes.  Same inputs as badbound.py.  Analysis infrastructure."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab
lab.build(); lab._init("h40a21")
cost = np.load('gpurun_out/times_ub_cost.npy'); bt = np.load('gpurun_out/times_ub_bt.npy')
idx = [i for i in range(1000) if bt[i] == 39]
variants = {
    "now: hard b1800 cap300": [dict(band=1800.0, cap=300, hs=1), dict(band=7200.0, cap=300, hs=0)],
    "cap450": [dict(band=1800.0, cap=450, hs=1), dict(band=7200.0, cap=450, hs=0)],
    "cap600": [dict(band=1800.0, cap=600, hs=1), dict(band=7200.0, cap=600, hs=0)],
    "b2700 cap450": [dict(band=2700.0, cap=450, hs=1), dict(band=7200.0, cap=450, hs=0)],
}
for name, atts in variants.items():
    rel = []; nodes = 0; second = 0
    for i in idx:
        g, v0, a0 = lab.grid_of(i)
        done = False
        for k, kw in enumerate(atts):
            o = lab.run_pass(g, v0, a0, **kw)
            nodes += o.nodes
            if o.complete:
                rel.append(o.cost / cost[i] - 1); done = True; second += k; break
        if not done: rel.append(np.inf)
    rel = np.array(rel)
    print("%-24s pre-pass nodes/episode %6.0f  second attempts %4d  unbounded %3d  bad(>20%%) %3d  >2%% %4d  below answer %3d" % (name, nodes / len(idx), second, np.isinf(rel).sum(), (rel > 0.2).sum(), (rel > 0.02).sum(), (rel < -1e-12).sum()))
