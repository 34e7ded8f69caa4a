"""ctypes wrapper of ``nj_oracle.c`` (the reference's no-jerk lattice solvers).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libnj_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "nj_oracle.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.run(["make", "-C", HERE, "-s", "libnj_oracle.so"], check=True)
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        L.orc_solve_no_jerk.argtypes = [C.c_int, C.POINTER(C.c_uint8), dp, C.c_int, dp, C.c_int, C.c_double, dp, dp, C.POINTER(C.c_longlong)]
        _lib = L
    return _lib


def solve_no_jerk(variant, obstacles, s_values, t_values, ego_start_speed, distances):
    """variant "fast" = st_cy.solve_s_t_path_no_jerk_fast, "djikstra" = st_cy.solve_s_t_path_no_jerk_djikstra."""
    ob = np.ascontiguousarray(obstacles).view(np.uint8)
    sv = np.ascontiguousarray(s_values, dtype=np.float64)
    tv = np.ascontiguousarray(t_values, dtype=np.float64)
    di = np.ascontiguousarray(distances, dtype=np.float64)
    seq = np.zeros(tv.size)
    pops = C.c_longlong(0)
    dp = C.POINTER(C.c_double)
    rc = lib().orc_solve_no_jerk(1 if variant == "djikstra" else 0, ob.ctypes.data_as(C.POINTER(C.c_uint8)), sv.ctypes.data_as(dp), sv.size,
                                 tv.ctypes.data_as(dp), tv.size, float(ego_start_speed), di.ctypes.data_as(dp), seq.ctypes.data_as(dp), C.byref(pops))
    if rc != 0:
        raise IndexError("index out of bounds in the seeding loop (the reference raises here too)")
    return seq, pops.value
