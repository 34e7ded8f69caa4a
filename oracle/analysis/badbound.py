"""Poor pre-pass bounds on the benchmark lattice: five offline studies behind one entry (round 2/3 analysis infrastructure, kept for the numbers
EXPERIMENTS.md quotes; nothing here is on the product path).  Inputs: gpurun_out/times_ub.bin, times_ub_cost.npy, times_ub_bt.npy written on a
GPU box by  STMPC_LIB=variants/libstmpc_times.so python scripts/lab/ub_dump.py  (timing build: scripts/lab/mk_times.py).
usage: python oracle/analysis/badbound.py which|aggregate|layers|detector|midcourse"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab


def study_which():
    """Episodes whose GPU pre-pass bound was far above the answer: which pre-pass design finds a usable bound?
    Inputs: gpurun_out/times_ub.bin, times_ub_cost.npy, times_ub_bt.npy written on a GPU box by
      STMPC_LIB=variants/libstmpc_times.so python scripts/lab/ub_dump.py   (timing build: scripts/lab/mk_times.py).  Analysis infrastructure."""
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


def study_aggregate():
    """Aggregate quality of pre-pass designs over the first 1000 benchmark states: bounds more than 20 % / 2 % above the answer, bounds below it
    (each costs a repeated exact pass), pre-pass nodes.  Same inputs as badbound.py.  Analysis infrastructure."""
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


def study_layers():
    """Layer by layer: where the reference answer's path drops out of the banded pre-pass (cost gap to the layer's cheapest node against the band in
    force).  Same inputs as badbound.py.  Analysis infrastructure."""
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


def study_detector():
    """Does the growth of the layer minimum flag a poor pre-pass bound?  Same inputs as badbound.py.  Analysis infrastructure."""
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


def study_midcourse():
    """Mid-course tightening: the exact pass (bounded by the pre-pass's U1) up to layer tc, then a banded completion from that layer's nodes.
    How many poor bounds does the completion repair, and what does it cost?  Same inputs as badbound.py.  Analysis infrastructure."""
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


STUDIES = {"which": study_which, "aggregate": study_aggregate, "layers": study_layers, "detector": study_detector, "midcourse": study_midcourse}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in STUDIES:
        sys.exit("usage: badbound.py " + "|".join(STUDIES))
    STUDIES[sys.argv[1]]()
