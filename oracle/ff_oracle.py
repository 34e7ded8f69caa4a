"""ctypes wrapper of the finer_fit oracle (``ff_oracle.c``).  TEST INFRASTRUCTURE ONLY (see st_oracle.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libff_oracle.so")
FF_NMAX = 64


class FFSettings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("v_max", "a_max", "a_min", "j_max", "j_min", "car_length")]


class FFQp(C.Structure):
    _fields_ = [("n", C.c_int), ("b", C.c_double * FF_NMAX), ("beq", C.c_double), ("cv", C.c_double),
                ("ca1", C.c_double), ("ca2", C.c_double), ("cj1", C.c_double), ("cj2", C.c_double), ("cj3", C.c_double),
                ("hV2", C.c_double), ("hA3_0", C.c_double), ("hA3", C.c_double), ("hA4_0", C.c_double), ("hA4", C.c_double),
                ("hJ5_0", C.c_double), ("hJ5_1", C.c_double), ("hJ5", C.c_double), ("hJ6_0", C.c_double),
                ("hJ6_1", C.c_double), ("hJ6", C.c_double), ("has_lo", C.c_int32 * FF_NMAX), ("has_hi", C.c_int32 * FF_NMAX),
                ("lo_h", C.c_double * FF_NMAX), ("hi_h", C.c_double * FF_NMAX)]


def build(force=False):
    src = os.path.join(HERE, "ff_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-s", "-B", "libff_oracle.so"], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        L.ff_sub_length.argtypes = [C.c_int, C.c_double, C.c_double]
        L.ff_build.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, C.POINTER(FFSettings), C.POINTER(FFQp)]
        L.ff_rows.argtypes = [C.POINTER(FFQp)]
        L.ff_dense.argtypes = [C.POINTER(FFQp), dp, dp, dp]
        L.ff_coneqp.argtypes = [C.POINTER(FFQp), C.c_int, dp, C.POINTER(C.c_int)]
        L.ff_coneqp_tol.argtypes = [C.POINTER(FFQp), C.c_int, dp, dp, C.POINTER(C.c_int)]
        L.ff_finer_fit.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, C.POINTER(FFSettings),
                                   C.c_int, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def settings(v_max, a_max, a_min, j_max, j_min, car_length):
    return FFSettings(float(v_max), float(a_max), float(a_min), float(j_max), float(j_min), float(car_length))


def build_qp(s_seq, dt, cdt, v0, a0, S, bac=None):
    s_seq = np.ascontiguousarray(s_seq, dtype=np.float64)
    q = FFQp()
    b = None if bac is None else np.ascontiguousarray(bac, dtype=np.float64)
    n = lib().ff_build(_dp(s_seq), len(s_seq), float(dt), float(cdt), float(v0), float(a0), None if b is None else _dp(b),
                       C.byref(S), C.byref(q))
    if n < 0:
        raise ValueError("ff_build failed: %d" % n)
    return q


def dense(q):
    n, m = q.n, lib().ff_rows(C.byref(q))
    G = np.zeros((m, n)); h = np.zeros(m); qv = np.zeros(n)
    lib().ff_dense(C.byref(q), _dp(G), _dp(h), _dp(qv))
    return G, h, qv


def coneqp(q, maxiters=10, tol=None):
    """tol = (abstol, reltol, feastol); None = cvxopt's defaults."""
    x = np.zeros(FF_NMAX); st = C.c_int(0)
    if tol is None:
        it = lib().ff_coneqp(C.byref(q), int(maxiters), _dp(x), C.byref(st))
    else:
        t = np.array(tol, dtype=np.float64)
        it = lib().ff_coneqp_tol(C.byref(q), int(maxiters), _dp(t), _dp(x), C.byref(st))
    return x[:q.n].copy(), it, st.value


def finer_fit(s_seq, dt, cdt, v0, a0, S, bac=None, maxiters=10):
    """st.finer_fit (st.py:584-723). Returns (x, iterations, status)."""
    s_seq = np.ascontiguousarray(s_seq, dtype=np.float64)
    out = np.zeros(FF_NMAX); it = C.c_int(0); st = C.c_int(0)
    b = None if bac is None else np.ascontiguousarray(bac, dtype=np.float64)
    n = lib().ff_finer_fit(_dp(s_seq), len(s_seq), float(dt), float(cdt), float(v0), float(a0), None if b is None else _dp(b),
                           C.byref(S), int(maxiters), _dp(out), C.byref(it), C.byref(st))
    if n < 0:
        raise ValueError("ff_finer_fit failed: %d" % n)
    return out[:n].copy(), it.value, st.value
