"""Mid-course tightening: the exact pass (bounded by the pre-pass's U1) up to layer tc, then a banded completion from that layer's nodes.
How many poor bounds does the completion repair, and what does it cost?  Same inputs as badbound.py.  Analysis infrastructure."""
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
bad = np.nonzero(rel > 0.2)[0][:80]
good = np.nonzero(ok & (rel <= 0.02))[0][:80]
for name, sel in (("bad", bad), ("good", good)):
    for tc in (8, 12, 16, 20):
        for cap in (300, 600):
            fixed = 0; comp_nodes = []; ex_nodes = []; r2 = []
            for i in sel:
                g, v0, a0 = lab.grid_of(int(i))
                o = lab.run_pass(g, v0, a0, U=float(U[i]), band=1800.0, cap=cap, hs=0, switch_t=tc)
                comp_nodes.append(o.tspan_over); ex_nodes.append(o.nodes - o.tspan_over)
                r = (o.cost / cost[i] - 1) if o.complete else np.inf
                r2.append(min(r, rel[i]))
            r2 = np.array(r2)
            print("%-4s tc=%2d cap=%3d: within 2%%: %2d  within 20%%: %2d of %d | exact nodes to tc: median %5d, completion nodes: median %5d" % (name, tc, cap, (r2 < 0.02).sum(), (r2 < 0.2).sum(), len(sel), np.median(ex_nodes), np.median(comp_nodes)))
