"""ctypes wrapper of the CPU oracle (``st_oracle.c``).

TEST INFRASTRUCTURE ONLY -- importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg; never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libst_oracle.so")
ORC_KMAX = 32

PARAM_FIELDS = ("future_s", "ds", "dt", "future_t", "start_unc", "unc_per_s",
                "d_w", "v_w", "a_w", "j_w", "v_des", "v_max", "a_min", "a_max", "j_min", "j_max", "min_allowed",
                "car_length", "crash_min_s", "max_pred_decel", "follow_gap", "react_thr", "crash_thr", "comb_min_dist")


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in PARAM_FIELDS]

    @classmethod
    def from_dict(cls, d):
        return cls(**{n: float(d[n]) for n in PARAM_FIELDS})


class OrcState(C.Structure):
    _fields_ = [("ego_x", C.c_double), ("ego_y", C.c_double), ("ego_v", C.c_double), ("ego_a", C.c_double),
                ("k", C.c_int), ("xs", C.c_double * ORC_KMAX), ("vs", C.c_double * ORC_KMAX)]


class OrcStats(C.Structure):
    _fields_ = [("nodes", C.c_longlong), ("edges", C.c_longlong), ("cells", C.c_longlong)]


def build(force=False):
    src = os.path.join(HERE, "st_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-s", "-B"], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        dp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        pp, sp = C.POINTER(OrcParams), C.POINTER(OrcState)
        L.orc_ego_s.argtypes = [C.c_double, C.c_double]; L.orc_ego_s.restype = C.c_double
        L.orc_num_s.argtypes = [pp, C.c_double]
        L.orc_num_t.argtypes = [pp]
        L.orc_predict_with_ego.argtypes = [pp, sp, C.c_double, C.c_double, C.c_double, sp]
        L.orc_predict_without_ego.argtypes = [pp, sp, C.c_double, C.c_double, sp]
        L.orc_build_grid.argtypes = [pp, sp, C.c_double, C.c_int, C.c_int, u8p, dp, dp, dp, dp]
        solver_args = [u8p, dp, C.c_int, dp, C.c_int, C.c_double, C.c_double, dp] + [C.c_double] * 11 + \
                      [dp, ip, ip, dp, C.POINTER(OrcStats)]
        L.orc_solve_heap.argtypes = solver_args
        L.orc_solve_layered.argtypes = solver_args
        L.orc_replay_cost.argtypes = [ip, C.c_int, dp, C.c_int, dp, C.c_double, C.c_double, dp, pp]
        L.orc_replay_cost.restype = C.c_double
        L.orc_solve_batch.argtypes = [pp, C.c_int, C.c_int, dp, ip, dp, dp, C.c_int, C.c_int, ip, ip, dp, dp, ip,
                                      C.POINTER(C.c_longlong)]
        L.orc_path_mean_abs_jerk.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double]
        L.orc_path_mean_abs_jerk.restype = C.c_double
        L.orc_pow3.argtypes = [C.c_double]; L.orc_pow3.restype = C.c_double
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def make_state(ego_x, ego_y, ego_v, ego_a, xs, vs):
    st = OrcState()
    st.ego_x, st.ego_y, st.ego_v, st.ego_a = float(ego_x), float(ego_y), float(ego_v), float(ego_a)
    st.k = len(xs)
    for i, (x, v) in enumerate(zip(xs, vs)):
        st.xs[i] = float(x)
        st.vs[i] = float(v)
    return st


def state_lists(st):
    return [st.xs[i] for i in range(st.k)], [st.vs[i] for i in range(st.k)]


def ego_s(x, y):
    return lib().orc_ego_s(float(x), float(y))


def predict_with_ego(params, st, selected_speed, dt, min_crash_distance=5.0):
    out = OrcState()
    crashed = lib().orc_predict_with_ego(C.byref(params), C.byref(st), float(selected_speed), float(dt),
                                         float(min_crash_distance), C.byref(out))
    return out, bool(crashed)


def predict_without_ego(params, st, dt, min_crash_distance=5.0):
    out = OrcState()
    crashed = lib().orc_predict_without_ego(C.byref(params), C.byref(st), float(dt), float(min_crash_distance),
                                            C.byref(out))
    return out, bool(crashed)


def build_grid(params, st, start_s, with_obs_tab=False):
    L = lib()
    S, H = L.orc_num_s(C.byref(params), float(start_s)), L.orc_num_t(C.byref(params))
    ob = np.zeros((H, S), dtype=np.uint8)
    di = np.zeros((H, S), dtype=np.float64)
    sv = np.zeros(S)
    tv = np.zeros(H)
    tab = np.zeros((H, ORC_KMAX)) if with_obs_tab else None
    L.orc_build_grid(C.byref(params), C.byref(st), float(start_s), S, H, ob.ctypes.data_as(C.POINTER(C.c_uint8)),
                     _dp(di), _dp(sv), _dp(tv), _dp(tab))
    if with_obs_tab:
        return ob.view(np.bool_), sv, tv, di, tab
    return ob.view(np.bool_), sv, tv, di


def solve_grid(obstacles, s_values, t_values, v0, a0, distances, tunables11, solver="heap"):
    """st_cy.solve_s_t_path_fast semantics. Returns (s_sequence, path_idx, best_t, cost, stats)."""
    L = lib()
    ob = np.ascontiguousarray(obstacles).view(np.uint8)
    sv = np.ascontiguousarray(s_values, dtype=np.float64)
    tv = np.ascontiguousarray(t_values, dtype=np.float64)
    di = np.ascontiguousarray(distances, dtype=np.float64)
    H, S = tv.shape[0], sv.shape[0]
    seq = np.zeros(H)
    pidx = np.zeros(H, dtype=np.int32)
    bt = C.c_int32(0)
    cost = C.c_double(0.0)
    stats = OrcStats()
    fn = L.orc_solve_heap if solver == "heap" else L.orc_solve_layered
    fn(ob.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(sv), S, _dp(tv), H, float(v0), float(a0), _dp(di),
       *[float(x) for x in tunables11], _dp(seq), _ip(pidx), C.byref(bt), C.byref(cost), C.byref(stats))
    return seq, pidx, bt.value, cost.value, {"nodes": stats.nodes, "edges": stats.edges, "cells": stats.cells}


def tunables_from_params(p):
    return (p.d_w, p.v_w, p.a_w, p.j_w, p.v_des, p.v_max, p.a_min, p.a_max, p.j_min, p.j_max, p.min_allowed)


def solve_batch(params, ego, k_count, other_x, other_v, solver="layered", nthreads=1):
    """Batched pipeline state -> path (st.py:726-754 + 790-802). Returns dict like the product's."""
    L = lib()
    ego = np.ascontiguousarray(ego, dtype=np.float64)
    N = ego.shape[0]
    k_count = np.ascontiguousarray(k_count, dtype=np.int32)
    ox = np.ascontiguousarray(other_x, dtype=np.float64).reshape(N, -1)
    ov = np.ascontiguousarray(other_v, dtype=np.float64).reshape(N, -1)
    Kmax = ox.shape[1]
    H = L.orc_num_t(C.byref(params))
    path = np.zeros((N, H), dtype=np.int32)
    bt = np.zeros(N, dtype=np.int32)
    cost = np.zeros(N)
    pd = np.zeros((N, H))
    crash = np.zeros(N, dtype=np.int32)
    counters = (C.c_longlong * 3)()
    rc = L.orc_solve_batch(C.byref(params), N, Kmax, _dp(ego), _ip(k_count), _dp(ox), _dp(ov),
                           0 if solver == "heap" else 1, int(nthreads), _ip(path), _ip(bt), _dp(cost), _dp(pd),
                           _ip(crash), counters)
    if rc != 0:
        raise RuntimeError("orc_solve_batch failed")
    return {"path_idx": path, "best_t": bt, "cost": cost, "path_dist": pd, "crash": crash,
            "nodes": counters[0], "edges": counters[1], "cells": counters[2]}
