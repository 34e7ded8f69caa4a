"""Offline study (analysis infrastructure; VERDICT r4 item 2 ii): how much of a FAILED exact pass could a retry reuse?
A pass under the bound U and its retry under U2 = 1.05 U (the first step of the retry ladder) are identical up to the first layer that holds a node
with U < cost <= U2: the failed pass dropped it, the retry expands it (also for a gentler step, 1.002).  For benchmark states with a complete path, with the bound set a hair BELOW
the reference's answer (what makes a pass fail on the GPU: bounds 0.00-0.2 % below C*, EXPERIMENTS.md round 5): the layer of that first node, and
the share of the retry's expanded nodes that lie before it.  usage: python oracle/analysis/retry_depth.py [n_states] [seed]"""
import ctypes as C
import sys

import numpy as np

import astar_lab as A
from astar_lab import L, Out, dp, orc, pkg, synth, _capi


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    H = _capi.num_t(p)
    ego, kc, ox, ov = synth.generate_states(4096, k=6, kmax=8, seed=seed)
    sel = np.arange(0, 4096, 4096 // n)[:n]
    ref = orc.solve_batch(op, ego[sel], kc[sel], ox[sel], ov[sel], solver="layered", nthreads=8)
    d = p.as_dict()
    F = np.zeros(8)
    u2 = C.c_double.in_dll(L, "astar_U2")
    for below, step in ((1e-5, 1.05), (1e-3, 1.05), (1e-5, 1.002), (1e-3, 1.002)):
        layers, share, nodes_fail, nodes_retry = [], [], [], []
        for j, i in enumerate(sel):
            if ref["best_t"][j] != H - 1:
                continue
            st_ = orc.make_state(*ego[i, :4], ox[i, :kc[i]], ov[i, :kc[i]])
            ob, sv, tv, di = orc.build_grid(op, st_, ego[i, 4])
            Cs = float(ref["cost"][j])
            U = Cs * (1.0 - below)

            def run(bound, U2):
                u2.value = U2
                out = Out(); path = np.zeros(H, dtype=np.int32)
                L.astar_pass(np.ascontiguousarray(ob).view(np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)), sv.ctypes.data_as(dp), sv.size, tv.ctypes.data_as(dp), H,
                             C.c_double(ego[i, 2]), C.c_double(ego[i, 3]), di.ctypes.data_as(dp), C.c_double(d["d_w"]), C.c_double(d["v_w"]), C.c_double(d["a_w"]),
                             C.c_double(d["j_w"]), C.c_double(d["v_des"]), C.c_double(d["v_max"]), C.c_double(d["a_min"]), C.c_double(d["a_max"]), C.c_double(d["j_min"]),
                             C.c_double(d["j_max"]), C.c_double(d["min_allowed"]), C.c_double(bound), 0, F.ctypes.data_as(dp), 0, 0, C.c_double(1.0),
                             path.ctypes.data_as(C.POINTER(C.c_int)), C.byref(out))
                return out
            fail = run(U, step * U)
            retry = run(step * U, 0.0)
            if not (retry.best_t == H - 1 and retry.cost == Cs):
                continue                      # (the retry fails too: bound still below the answer)
            if fail.first_cut_layer < 0:
                continue
            layers.append(fail.first_cut_layer); share.append(fail.nodes_before_cut / max(retry.nodes, 1))
            nodes_fail.append(fail.nodes); nodes_retry.append(retry.nodes)
        layers, share = np.array(layers), np.array(share)
        print("bound %.0e below C*, retry at %.3f U: %d states; first layer (of %d) with a node in (U, U2]: quantiles 10/25/50/75/90 %% = %s; share of the retry's nodes that lie before it: "
              "mean %.3f, quantiles %s; nodes failed pass %.0f, retry %.0f per state" % (below, step, layers.size, H - 1, np.quantile(layers, [.1, .25, .5, .75, .9]).astype(int),
                                                                                         share.mean(), np.round(np.quantile(share, [.1, .25, .5, .75, .9]), 3), np.mean(nodes_fail), np.mean(nodes_retry)))


if __name__ == "__main__":
    main()
