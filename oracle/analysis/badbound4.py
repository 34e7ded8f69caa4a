"""Layer by layer: where the reference answer's path drops out of the banded pre-pass (cost gap to the layer's cheapest node against the band in
force).  Same inputs as badbound.py.  Analysis infrastructure."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab
from oracle import st_oracle as orc
lab.build(); lab._init("h40a21")
cost = np.load('gpurun_out/times_ub_cost.npy'); bt = np.load('gpurun_out/times_ub_bt.npy')
ego, k, ox, ov = lab._g["states"]
shown = 0
for i in range(600):
    if bt[i] != 39: continue
    g, v0, a0 = lab.grid_of(i)
    o = lab.run_pass(g, v0, a0, band=1800.0, cap=300, hs=1)
    if not o.complete or o.cost / cost[i] - 1 < 0.2: continue
    ref = orc.solve_batch(lab._g["op"], ego[i:i+1], k[i:i+1], ox[i:i+1], ov[i:i+1], solver="layered", nthreads=1)
    path = np.ascontiguousarray(ref["path_idx"][0].astype(np.int32))
    lab._g["L"].lab_set_watch(path.ctypes.data_as(C.POINTER(C.c_int)))
    o = lab.run_pass(g, v0, a0, band=1800.0, cap=300, hs=1)
    lab._g["L"].lab_set_watch(None)
    print("episode %d: answer %.0f, pre-pass %.0f; ego v0 %.1f a0 %.1f" % (i, cost[i], o.cost, v0, a0))
    print("   t: nodes | layer min | band | answer-path cell: cost there, selected?")
    for t in range(0, 39):
        flag = "" if o.watch_sel[t] else "   <-- not expanded"
        print("  %2d: %4d | %9.1f | %7.1f | cell %5d cost %10.1f (gap %8.1f)%s" % (t, o.per_layer[t], o.lay_kmin[t], o.lay_band[t], path[t], o.watch_c[t], o.watch_c[t] - o.lay_kmin[t], flag))
        if not o.watch_sel[t]: break
    shown += 1
    if shown >= 4: break
