"""Offline study (analysis infrastructure): the bounded exact pass with and without a cost-to-go lower bound (astar_lab.c) on benchmark states.
usage: python oracle/analysis/astar_lab.py [n_states] [seed]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, synth
from oracle import st_oracle as orc

so = os.path.join(HERE, "libastar_lab.so")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(HERE, "astar_lab.c")), os.path.getmtime(os.path.join(HERE, "..", "st_oracle.c"))):
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", os.path.join(HERE, "astar_lab.c"), "-o", so, "-lm"], check=True)
L = C.CDLL(so)


class Out(C.Structure):
    _fields_ = [("nodes", C.c_longlong), ("edges", C.c_longlong), ("edges_filt", C.c_longlong), ("cut_h", C.c_longlong), ("reached", C.c_longlong),
                ("best_t", C.c_int), ("cost", C.c_double), ("pruned", C.c_int), ("span_sum", C.c_longlong), ("layers", C.c_longlong),
                ("first_cut_layer", C.c_int), ("nodes_before_cut", C.c_longlong)]


dp = C.POINTER(C.c_double)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    H = _capi.num_t(p)
    ego, kc, ox, ov = synth.generate_states(4096, k=6, kmax=8, seed=seed)
    sel = np.arange(0, 4096, 4096 // n)[:n]
    ref = orc.solve_batch(op, ego[sel], kc[sel], ox[sel], ov[sel], solver="layered", nthreads=8)
    d = p.as_dict()
    ds, dt = d["ds"], d["dt"]
    imax = int(np.floor(d["v_max"] * dt / ds + 1e-9))
    D = int(np.ceil(max(abs(d["a_min"]), abs(d["a_max"])) * dt * dt / ds)) + 2
    nd = 2 * D + 1
    F = np.zeros((H, imax + 1, nd))
    L.astar_table(C.c_double(ds), C.c_double(dt), H, C.c_double(d["v_w"]), C.c_double(d["a_w"]), C.c_double(d["j_w"]), C.c_double(d["v_des"]), C.c_double(d["v_max"]),
                  C.c_double(d["a_min"]), C.c_double(d["a_max"]), C.c_double(d["j_min"]), C.c_double(d["j_max"]), C.c_double(1e-6), imax, D, F.ctypes.data_as(dp))
    print("table: imax %d D %d, F[H-1] at standstill %.1f, at 15 m/s steady %.1f, at 30 m/s steady %.1f" % (imax, D, F[H - 1, 0, D], F[H - 1, 90, D], F[H - 1, 180, D]))
    tot = {}
    bad = 0
    badk = {}
    why = []
    for j, i in enumerate(sel):
        if ref["best_t"][j] != H - 1:
            continue
        st_ = orc.make_state(*ego[i, :4], ox[i, :kc[i]], ov[i, :kc[i]])
        ob, sv, tv, di = orc.build_grid(op, st_, ego[i, 4])
        S = sv.size
        Cs = float(ref["cost"][j])
        for tag, mult in (("1.00002", 1.00002), ("1.015", 1.015), ("1.2", 1.2)):
            for use_h in (0, 1, 2):
                out = Out()
                path = np.zeros(H, dtype=np.int32)
                L.astar_pass(np.ascontiguousarray(ob).view(np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)), sv.ctypes.data_as(dp), S, tv.ctypes.data_as(dp), H,
                             C.c_double(ego[i, 2]), C.c_double(ego[i, 3]), di.ctypes.data_as(dp), C.c_double(d["d_w"]), C.c_double(d["v_w"]), C.c_double(d["a_w"]),
                             C.c_double(d["j_w"]), C.c_double(d["v_des"]), C.c_double(d["v_max"]), C.c_double(d["a_min"]), C.c_double(d["a_max"]), C.c_double(d["j_min"]),
                             C.c_double(d["j_max"]), C.c_double(d["min_allowed"]), C.c_double(Cs * mult), use_h, F.ctypes.data_as(dp), imax, D, C.c_double(1.0 - 1e-9),
                             path.ctypes.data_as(C.POINTER(C.c_int)), C.byref(out))
                ok = out.best_t == H - 1 and out.cost == Cs and np.array_equal(path, ref["path_idx"][j])
                bad += 0 if ok else 1
                if not ok: badk[(tag, use_h)] = badk.get((tag, use_h), 0) + 1; why.append((int(i), tag, use_h, out.best_t, out.cost - Cs, int((path != ref['path_idx'][j]).sum())))
                key = (tag, use_h)
                t_ = tot.setdefault(key, [0, 0, 0, 0])
                t_[0] += out.nodes; t_[1] += out.edges; t_[2] += out.edges_filt; t_[3] += 1
    print("differing by variant:", badk, why[:12])
    print("states with a complete path: %d; results that differ from the oracle: %d" % (tot[("1.00002", 0)][3], bad))
    for (tag, use_h), (nodes, edges, ef, cnt) in sorted(tot.items()):
        print("U = %-8s C*  %s: nodes %8.0f  edges %9.0f  candidates after the quadratic filter %9.0f   per solve" % (
            tag, ["plain bound        ", "+ cost-to-go (node)", "+ cost-to-go (cand)"][use_h], nodes / cnt, edges / cnt, ef / cnt))


if __name__ == "__main__":
    main()
