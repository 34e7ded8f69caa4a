"""ctypes binding of the C-ABI declared in ``include/stmpc.h``.

This is the whole Python<->native boundary: plain pointers and sizes.  There is no CPU
fallback -- if ``libstmpc.so`` is missing the import of this module raises, and if no GPU
is visible ``Context()`` raises ``StmpcError(STMPC_ENODEV)``.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

STMPC_OK, STMPC_EINVAL, STMPC_ENODEV, STMPC_EHIP, STMPC_ENOMEM, STMPC_EINTERNAL = 0, -1, -2, -3, -4, -5
KMAX_LIMIT, H_LIMIT, S_LIMIT = 32, 64, 65000


class StmpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("stmpc error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    """``stmpc_params`` (include/stmpc.h)."""
    _fields_ = [(n, C.c_double) for n in (
        "future_s", "ds", "dt", "future_t", "start_unc", "unc_per_s",
        "d_w", "v_w", "a_w", "j_w", "v_des", "v_max", "a_min", "a_max", "j_min", "j_max", "min_allowed",
        "car_length", "crash_min_s", "max_pred_decel", "follow_gap", "react_thr", "crash_thr", "comb_min_dist")]

    @classmethod
    def from_settings(cls, S):
        """Read the flags exactly where the reference reads them (st.py:727-746, 37, 46, 800;
        prediction.py:11-12, 85-86)."""
        return cls(
            future_s=S.FUTURE_S, ds=S.S_DISCRETIZATION, dt=S.T_DISCRETIZATION, future_t=S.FUTURE_T,
            start_unc=S.START_UNCERTAINTY, unc_per_s=S.UNCERTAINTY_PER_SECOND,
            d_w=S.D_WEIGHT, v_w=S.V_WEIGHT, a_w=S.A_WEIGHT, j_w=S.J_WEIGHT, v_des=S.DESIRED_SPEED,
            v_max=S.MAX_SPEED, a_min=S.MAX_NEGATIVE_ACCELERATION, a_max=S.MAX_POSITIVE_ACCELERATION,
            j_min=S.MINIMUM_NEGATIVE_JERK, j_max=S.MAXIMUM_POSITIVE_JERK, min_allowed=S.MIN_ALLOWED_DISTANCE,
            car_length=S.CAR_LENGTH, crash_min_s=S.CRASH_MIN_S, max_pred_decel=S.MAX_PREDICTED_DECELERATION,
            follow_gap=30.0, react_thr=8.0, crash_thr=11.0, comb_min_dist=S.COMBINATION_MIN_DISTANCE)

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [("episodes", C.c_int64), ("fast_path", C.c_int64), ("fallback", C.c_int64), ("hbm_tier", C.c_int64),
                ("retries", C.c_int64), ("nodes_exact", C.c_int64), ("nodes_bound", C.c_int64),
                ("solve_ms", C.c_double), ("dp_kernel_ms", C.c_double), ("guided", C.c_int64), ("resume_refused", C.c_int64), ("pool_exhausted", C.c_int64)]


class CombinedCfg(C.Structure):
    """``stmpc_combined_cfg`` (include/stmpc.h): the flags of dqn.RLAgent.do_combined_control."""
    _fields_ = [("tick_length", C.c_double), ("stop_x", C.c_double), ("rollout_length", C.c_int32), ("st_test_rollouts", C.c_int32),
                ("check_rollout_crash", C.c_int32), ("limit_dqn_speed", C.c_int32), ("test_rollout_state", C.c_int32),
                ("test_st_strictly_better", C.c_int32), ("remember_last_choice", C.c_int32), ("sparse_control", C.c_int32)]

    @classmethod
    def from_settings(cls, S, sparse_control=False):
        """``sparse_control`` (not a reference flag): False (the C-level default: a zeroed field) keeps ``stmpc_combined_decide_device`` fully asynchronous
        -- nothing leaves the GPU, the call can be captured in a hipGraph -- and solves st.do_st_control for every state; True solves it only for
        the states whose decision calls it, as the reference does, at the price of one host round trip per decide call (the count of those
        states).  ``EpisodeRunner`` and the benchmarks ask for it explicitly."""
        return cls(sparse_control=int(bool(sparse_control)), tick_length=S.TICK_LENGTH, stop_x=S.STOP_X, rollout_length=int(S.ROLLOUT_LENGTH), st_test_rollouts=int(S.ST_TEST_ROLLOUTS),
                   check_rollout_crash=int(bool(S.CHECK_ROLLOUT_CRASH)), limit_dqn_speed=int(bool(getattr(S, "LIMIT_DQN_SPEED", False))),
                   test_rollout_state=int(bool(S.TEST_ROLLOUT_STATE)), test_st_strictly_better=int(bool(getattr(S, "TEST_ST_STRICTLY_BETTER", False))),
                   remember_last_choice=int(bool(getattr(S, "REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED", False))))


class FeaturesCfg(C.Structure):
    """``stmpc_policy_features_cfg`` (include/stmpc.h): the flags of dqn.get_state_vector_from_base_state (+ the TimeFeature input of ddpg.py:41)."""
    _fields_ = [("max_speed", C.c_double), ("sensor_radius", C.c_double), ("time_scale", C.c_double), ("cars_ahead", C.c_int32), ("cars_behind", C.c_int32),
                ("use_acceleration", C.c_int32), ("use_speed_difference", C.c_int32), ("normalize", C.c_int32), ("time_feature", C.c_int32)]

    @classmethod
    def from_settings(cls, S, time_feature=True):
        g = lambda name, default: getattr(S, name, default)             # the reference's defaults (config.py:35,48-49,136-139)
        return cls(max_speed=float(S.MAX_SPEED), sensor_radius=float(g("SENSOR_RADIUS", 125)), time_scale=0.001, cars_ahead=int(g("CARS_AHEAD", 2)),
                   cars_behind=int(g("CARS_BEHIND", 2)), use_acceleration=int(bool(g("USE_ACCELERATION_OF_OTHER_CARS", True))),
                   use_speed_difference=int(bool(g("USE_SPEED_DIFFERENCE", True))), normalize=int(bool(g("NORMALIZE_VECTOR_INPUT", True))),
                   time_feature=int(bool(time_feature)))


class SimCfg(C.Structure):
    """``stmpc_sim_cfg`` (include/stmpc.h): scenario constants of the batched SUMO-free episode runner."""
    _fields_ = [("tick_length", C.c_double), ("other_car_speed", C.c_double), ("base_traffic_interval", C.c_double), ("spawn_x", C.c_double),
                ("despawn_x", C.c_double), ("ego_start_x", C.c_double), ("ego_start_y", C.c_double), ("arrive_x", C.c_double),
                ("sensor_radius", C.c_double), ("start_speed", C.c_double), ("start_speed_std", C.c_double), ("min_start_speed", C.c_double),
                ("max_start_speed", C.c_double), ("veh_accel", C.c_double), ("veh_decel", C.c_double), ("veh_min_gap", C.c_double), ("veh_tau", C.c_double),
                ("veh_emergency_decel", C.c_double), ("veh_length", C.c_double), ("veh_width", C.c_double), ("speed_dev", C.c_double),
                ("vary_traffic_start_times", C.c_int32), ("randomize_start_speed", C.c_int32),
                ("max_ticks", C.c_int32), ("yield_overlap", C.c_int32), ("seed", C.c_uint64),
                ("ego_route_xy", C.POINTER(C.c_double)), ("ego_route_n", C.c_int32), ("reserved0", C.c_int32),
                ("disruption_min_s", C.c_double)]

    def set_route(self, xy):
        """``xy``: [n][2] polyline of the ego's lane centre line (x strictly increasing), or None for the planner's straight lines.
        The array is kept alive by this object; ``stmpc_sim_init_device`` copies it to the device."""
        if xy is None:
            self._route = None
            self.ego_route_xy, self.ego_route_n = None, 0
        else:
            import numpy as np
            self._route = np.ascontiguousarray(xy, dtype=np.float64)
            assert self._route.ndim == 2 and self._route.shape[1] == 2
            self.ego_route_xy = self._route.ctypes.data_as(C.POINTER(C.c_double))
            self.ego_route_n = int(self._route.shape[0])
        return self


class ProfileTotals(C.Structure):
    _fields_ = [("launches", C.c_int64), ("episodes", C.c_int64), ("solve_ms", C.c_double), ("dp_kernel_ms", C.c_double)]


_lib = None

EXPORTS = (
    "stmpc_backend_info", "stmpc_last_error", "stmpc_create", "stmpc_destroy", "stmpc_ego_s", "stmpc_num_s",
    "stmpc_num_t", "stmpc_path_mean_abs_jerk", "stmpc_solve_batch_device", "stmpc_solve_batch", "stmpc_get_stats",
    "stmpc_solve_grid", "stmpc_build_grid", "stmpc_predict_batch", "stmpc_probe_arith", "stmpc_profile",
    "stmpc_finer_fit_batch", "stmpc_st_control_batch", "stmpc_st_control_batch_device",
    "stmpc_rollout_step_device", "stmpc_combined_decide_device", "stmpc_combined_read_state", "stmpc_solve_grid_no_jerk",
    "stmpc_sim_init_device", "stmpc_sim_view_device", "stmpc_sim_step_device", "stmpc_sim_read", "stmpc_fastdiv2_check",
    "stmpc_abi_version", "stmpc_check_error", "stmpc_predict_batch_acc", "stmpc_sim_status_device",
    "stmpc_policy_features_device", "stmpc_policy_features_len", "stmpc_combined_counts", "stmpc_solve_batch_device_ac",
    "stmpc_actor_create", "stmpc_actor_destroy", "stmpc_actor_eval_device",
)
SIM_NACC = 12        # STMPC_SIM_NACC
ABI_VERSION = 6     # STMPC_ABI_VERSION of include/stmpc.h this binding was written against

QP_NMAX = 64        # STMPC_QP_NMAX
QP_MAXITERS = 10    # STMPC_QP_MAXITERS (solvers.options['maxiters'], st.py:17)


def lib_path():
    # STMPC_LIB selects another build of the same ABI (A/B measurements of kernel variants)
    return os.environ.get("STMPC_LIB") or _build.LIB_PATH


def load():
    """Load ``libstmpc.so`` (must have been built in-tree: ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and must be the first to load it, otherwise the
    # system copy this library links against initialises the device and torch then reports "No HIP GPUs are available"
    # (and device pointers could not be shared).  With torch imported first the loader resolves this library's
    # libamdhip64 dependency to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(this package has no CPU fallback)" % path)
    lib = C.CDLL(path)
    have = lib.stmpc_abi_version() if hasattr(lib, "stmpc_abi_version") else 0
    if have != ABI_VERSION:
        raise ImportError("%s implements ABI %d, this binding expects %d: rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)"
                          % (path, have, ABI_VERSION))
    dp, ip, u8p, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_void_p
    pp = C.POINTER(Params)
    lib.stmpc_backend_info.restype = C.c_char_p
    lib.stmpc_last_error.restype = C.c_char_p
    lib.stmpc_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.stmpc_destroy.argtypes = [vp]
    lib.stmpc_destroy.restype = None
    lib.stmpc_ego_s.argtypes = [C.c_double, C.c_double]
    lib.stmpc_ego_s.restype = C.c_double
    lib.stmpc_num_s.argtypes = [pp, C.c_double]
    lib.stmpc_num_t.argtypes = [pp]
    lib.stmpc_path_mean_abs_jerk.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double]
    lib.stmpc_path_mean_abs_jerk.restype = C.c_double
    lib.stmpc_fastdiv2_check.argtypes = [C.c_double, C.POINTER(C.c_double)]
    lib.stmpc_solve_batch_device.argtypes = [vp, pp, C.c_int, C.c_int] + [vp] * 9 + [vp]
    lib.stmpc_solve_batch_device_ac.argtypes = [vp, pp, C.c_int, C.c_int] + [vp] * 10 + [vp]
    lib.stmpc_solve_batch.argtypes = [vp, pp, C.c_int, C.c_int, dp, ip, dp, dp, ip, ip, dp, dp, ip]
    lib.stmpc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.stmpc_solve_grid.argtypes = [vp, u8p, dp, C.c_int, dp, C.c_int, C.c_double, C.c_double, dp] + [C.c_double] * 11 + [dp]
    lib.stmpc_solve_grid_no_jerk.argtypes = [vp, C.c_int, u8p, dp, C.c_int, dp, C.c_int, C.c_double, dp, dp]
    lib.stmpc_build_grid.argtypes = [vp, pp, dp, C.c_int, dp, dp, u8p, dp, dp, dp]
    lib.stmpc_predict_batch.argtypes = [vp, pp, C.c_int, C.c_int, C.c_int, dp, ip, dp, dp, dp, C.c_double, C.c_double,
                                        dp, dp, dp, ip]
    lib.stmpc_predict_batch_acc.argtypes = [vp, pp, C.c_int, C.c_int, C.c_int, dp, ip, dp, dp, dp, C.c_double, C.c_double,
                                            dp, dp, dp, ip, dp]
    lib.stmpc_check_error.argtypes = [vp]
    lib.stmpc_sim_status_device.argtypes = [vp, C.c_int, vp, vp]
    cp = C.POINTER(CombinedCfg)
    lib.stmpc_rollout_step_device.argtypes = [vp, pp, cp, C.c_int, C.c_int, C.c_int] + [vp] * 7 + [vp]
    lib.stmpc_combined_decide_device.argtypes = [vp, pp, cp, C.c_int, C.c_int] + [vp] * 12 + [vp]
    sp = C.POINTER(SimCfg)
    lib.stmpc_policy_features_len.argtypes = [C.POINTER(FeaturesCfg)]
    lib.stmpc_policy_features_device.argtypes = [vp, C.POINTER(FeaturesCfg), C.c_int, C.c_int, C.c_int] + [vp] * 7 + [C.c_int, vp]
    fp = C.POINTER(C.c_float)
    lib.stmpc_actor_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, C.c_double, C.c_double, C.POINTER(vp)]
    lib.stmpc_actor_destroy.argtypes = [vp]
    lib.stmpc_actor_destroy.restype = None
    lib.stmpc_actor_eval_device.argtypes = [vp, vp, C.POINTER(FeaturesCfg), C.c_int, C.c_int, C.c_int] + [vp] * 7 + [C.c_int, vp, vp]
    lib.stmpc_combined_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
    lib.stmpc_sim_init_device.argtypes = [vp, sp, C.c_int, vp]
    lib.stmpc_sim_view_device.argtypes = [vp, sp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.stmpc_sim_step_device.argtypes = [vp, pp, sp, C.c_int, vp, vp]
    lib.stmpc_sim_read.argtypes = [vp, C.c_int, ip, ip, dp, dp]
    lib.stmpc_combined_read_state.argtypes = [vp, C.c_int, ip, ip, ip, dp, dp, ip, dp, dp, dp, ip, dp, dp, ip]
    lib.stmpc_probe_arith.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
    lib.stmpc_profile.argtypes = [vp, C.c_int, C.POINTER(ProfileTotals)]
    lib.stmpc_finer_fit_batch.argtypes = [vp, pp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, dp, ip, dp, dp, dp,
                                          C.c_int, dp, ip, ip]
    lib.stmpc_st_control_batch.argtypes = [vp, pp, C.c_double, C.c_int, C.c_int, dp, ip, dp, dp, dp, ip, ip, dp, dp, ip]
    lib.stmpc_st_control_batch_device.argtypes = [vp, pp, C.c_double, C.c_int, C.c_int] + [vp] * 10 + [vp]
    _lib = lib
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _u8ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def ego_s(x, y):
    """control.get_ego_s evaluated by the library's host helper (same libm as CPython)."""
    return load().stmpc_ego_s(float(x), float(y))


def num_s(params, start_s):
    return load().stmpc_num_s(C.byref(params), float(start_s))


def num_t(params):
    return load().stmpc_num_t(C.byref(params))


def path_mean_abs_jerk(seq, v0, a0, dt):
    return load().stmpc_path_mean_abs_jerk(_dptr(seq), int(seq.size), float(v0), float(a0), float(dt))


def fastdiv2_check(d):
    """(ok, zl): whether the solver may form x / d as fma(x, RN(1/d), x * zl) -- the check it applies to dt, dt**2, dt**3."""
    zl = C.c_double(0.0)
    ok = load().stmpc_fastdiv2_check(float(d), C.byref(zl))
    return bool(ok), zl.value


def backend_info():
    return load().stmpc_backend_info().decode()


def library_source_hash():
    """sha256[:16] of the sources the loaded library was built from (``build.source_hash`` at build time), or "unknown"."""
    info = backend_info()
    return info.rsplit("src=", 1)[1].strip() if "src=" in info else "unknown"


class Context:
    """One HIP device + its scratch (``stmpc_ctx``)."""

    def __init__(self, device=-1):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.stmpc_create(C.byref(h), int(device))
        if rc != 0:
            raise StmpcError(rc, self._lib.stmpc_last_error().decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.stmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise StmpcError(rc, self._lib.stmpc_last_error().decode())

    # -- batched solve, host numpy arrays --------------------------------------------------
    def solve_batch(self, params, ego, k_count, other_x, other_v, want_dist=True):
        ego = np.ascontiguousarray(ego, dtype=np.float64)
        k_count = np.ascontiguousarray(k_count, dtype=np.int32)
        N = ego.shape[0]
        if ego.ndim != 2 or ego.shape[1] != 5:
            raise ValueError("ego must be [N,5] (x, y, v, a, start_s)")
        other_x = np.ascontiguousarray(other_x, dtype=np.float64)
        other_v = np.ascontiguousarray(other_v, dtype=np.float64)
        Kmax = other_x.shape[1] if other_x.ndim == 2 else (other_x.size // N if N else 0)
        other_x = other_x.reshape(N, Kmax)
        other_v = other_v.reshape(N, Kmax)
        H = num_t(params)
        path = np.empty((N, H), dtype=np.int32)
        best_t = np.empty(N, dtype=np.int32)
        cost = np.empty(N, dtype=np.float64)
        pdist = np.empty((N, H), dtype=np.float64) if want_dist else None
        crash = np.empty(N, dtype=np.int32) if want_dist else None
        self._chk(self._lib.stmpc_solve_batch(self._h, C.byref(params), N, Kmax, _dptr(ego), _iptr(k_count),
                                              _dptr(other_x) if Kmax else None, _dptr(other_v) if Kmax else None,
                                              _iptr(path), _iptr(best_t), _dptr(cost), _dptr(pdist), _iptr(crash)))
        return path, best_t, cost, pdist, crash

    # -- batched solve, device pointers (ints), asynchronous ---------------------------------
    def solve_batch_device(self, params, N, Kmax, d_ego, d_k, d_ox, d_ov, d_path, d_bt, d_cost, d_pd=0, d_crash=0,
                           stream=0, d_action_cost=0):
        """``d_action_cost``: optional fp64 [N][2] device buffer the solver fills with (first-step cell, cost) -- the fused row of the multi-GPU gather."""
        self._chk(self._lib.stmpc_solve_batch_device_ac(self._h, C.byref(params), int(N), int(Kmax), d_ego, d_k, d_ox,
                                                        d_ov, d_path, d_bt, d_cost, d_pd or None, d_crash or None,
                                                        d_action_cost or None, stream or None))

    # -- st.finer_fit, batched (host arrays) ------------------------------------------------------
    def finer_fit_batch(self, params, delta_t, coarse_delta_t, s_seq, lengths, v0, a0, bac=None, maxiters=QP_MAXITERS):
        """Returns ``(out[N, QP_NMAX], out_len[N], iters[N])``; see ``stmpc_finer_fit_batch`` in include/stmpc.h."""
        s_seq = np.ascontiguousarray(s_seq, dtype=np.float64)
        if s_seq.ndim != 2:
            raise ValueError("s_seq must be [N, Hs]")
        N, Hs = s_seq.shape
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        v0 = np.ascontiguousarray(v0, dtype=np.float64)
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        if bac is not None:
            bac = np.ascontiguousarray(bac, dtype=np.float64).reshape(N, 4)
        out = np.zeros((N, QP_NMAX), dtype=np.float64)
        out_len = np.zeros(N, dtype=np.int32)
        iters = np.zeros(N, dtype=np.int32)
        self._chk(self._lib.stmpc_finer_fit_batch(self._h, C.byref(params), float(delta_t), float(coarse_delta_t),
                                                  int(maxiters), N, Hs, _dptr(s_seq), _iptr(lengths), _dptr(v0), _dptr(a0),
                                                  _dptr(bac), QP_NMAX, _dptr(out), _iptr(out_len), _iptr(iters)))
        return out, out_len, iters

    # -- st.do_st_control, batched (host arrays) ---------------------------------------------------
    def st_control_batch(self, params, tick_length, ego, k_count, other_x, other_v, want_paths=False):
        """Returns a dict: ``speed[N]``, ``best_t[N]`` and, with ``want_paths``, ``path_idx``, ``cost``, ``fine``, ``fine_len``."""
        ego = np.ascontiguousarray(ego, dtype=np.float64)
        k_count = np.ascontiguousarray(k_count, dtype=np.int32)
        N = ego.shape[0]
        if ego.ndim != 2 or ego.shape[1] != 5:
            raise ValueError("ego must be [N,5] (x, y, v, a, start_s)")
        other_x = np.ascontiguousarray(other_x, dtype=np.float64)
        other_v = np.ascontiguousarray(other_v, dtype=np.float64)
        Kmax = other_x.shape[1] if other_x.ndim == 2 else (other_x.size // N if N else 0)
        other_x = other_x.reshape(N, Kmax)
        other_v = other_v.reshape(N, Kmax)
        H = num_t(params)
        speed = np.zeros(N, dtype=np.float64)
        best_t = np.zeros(N, dtype=np.int32)
        path = np.empty((N, H), dtype=np.int32) if want_paths else None
        cost = np.empty(N, dtype=np.float64) if want_paths else None
        fine = np.zeros((N, QP_NMAX), dtype=np.float64) if want_paths else None
        fine_len = np.zeros(N, dtype=np.int32) if want_paths else None
        self._chk(self._lib.stmpc_st_control_batch(self._h, C.byref(params), float(tick_length), N, Kmax, _dptr(ego),
                                                   _iptr(k_count), _dptr(other_x) if Kmax else None,
                                                   _dptr(other_v) if Kmax else None, _dptr(speed), _iptr(best_t),
                                                   _iptr(path), _dptr(cost), _dptr(fine), _iptr(fine_len)))
        res = {"speed": speed, "best_t": best_t}
        if want_paths:
            res.update(path_idx=path, cost=cost, fine=fine, fine_len=fine_len)
        return res

    def st_control_batch_device(self, params, tick_length, N, Kmax, d_ego, d_k, d_ox, d_ov, d_path, d_bt, d_cost, d_speed,
                                d_fine=0, d_fine_len=0, stream=0):
        self._chk(self._lib.stmpc_st_control_batch_device(self._h, C.byref(params), float(tick_length), int(N), int(Kmax),
                                                          d_ego, d_k, d_ox, d_ov, d_path, d_bt, d_cost, d_speed,
                                                          d_fine or None, d_fine_len or None, stream or None))

    def profile_begin(self):
        self._chk(self._lib.stmpc_profile(self._h, 1, None))

    def profile_end(self):
        t = ProfileTotals()
        self._chk(self._lib.stmpc_profile(self._h, 0, C.byref(t)))
        return {n: getattr(t, n) for n, _ in ProfileTotals._fields_}

    def stats(self):
        s = Stats()
        self._chk(self._lib.stmpc_get_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in Stats._fields_}

    # -- st_cy.solve_s_t_path_fast semantics -------------------------------------------------------
    def solve_grid(self, obstacles, s_values, t_values, v0, a0, distances, tunables11):
        ob = np.ascontiguousarray(obstacles).view(np.uint8) if obstacles.dtype == np.bool_ else \
            np.ascontiguousarray(obstacles, dtype=np.uint8)
        sv = np.ascontiguousarray(s_values, dtype=np.float64)
        tv = np.ascontiguousarray(t_values, dtype=np.float64)
        di = np.ascontiguousarray(distances, dtype=np.float64)
        H, S = tv.shape[0], sv.shape[0]
        if ob.shape != (H, S) or di.shape != (H, S):
            raise ValueError("obstacles/distances must be [num_t, num_s]")
        out = np.empty(H, dtype=np.float64)
        self._chk(self._lib.stmpc_solve_grid(self._h, _u8ptr(ob), _dptr(sv), S, _dptr(tv), H, float(v0), float(a0),
                                             _dptr(di), *[float(x) for x in tunables11], _dptr(out)))
        return out

    def solve_grid_no_jerk(self, variant, obstacles, s_values, t_values, v0, distances):
        """variant 0 = st_cy.solve_s_t_path_no_jerk_fast, 1 = st_cy.solve_s_t_path_no_jerk_djikstra."""
        ob = np.ascontiguousarray(obstacles).view(np.uint8)
        sv = np.ascontiguousarray(s_values, dtype=np.float64)
        tv = np.ascontiguousarray(t_values, dtype=np.float64)
        di = np.ascontiguousarray(distances, dtype=np.float64)
        if ob.shape != (tv.size, sv.size) or di.shape != ob.shape:
            raise ValueError("obstacles / distances must be [num_t, num_s]")
        seq = np.zeros(tv.size)
        self._chk(self._lib.stmpc_solve_grid_no_jerk(self._h, int(variant), _u8ptr(ob), _dptr(sv), sv.size, _dptr(tv), tv.size, float(v0), _dptr(di), _dptr(seq)))
        return seq

    def build_grid(self, params, state5, other_x, other_v):
        state5 = np.ascontiguousarray(state5, dtype=np.float64)
        ox = np.ascontiguousarray(other_x, dtype=np.float64)
        ov = np.ascontiguousarray(other_v, dtype=np.float64)
        k = ox.shape[0]
        H, S = num_t(params), num_s(params, state5[4])
        ob = np.empty((H, S), dtype=np.uint8)
        di = np.empty((H, S), dtype=np.float64)
        sv = np.empty(S, dtype=np.float64)
        tv = np.empty(H, dtype=np.float64)
        self._chk(self._lib.stmpc_build_grid(self._h, C.byref(params), _dptr(state5), k, _dptr(ox) if k else None,
                                             _dptr(ov) if k else None, _u8ptr(ob), _dptr(di), _dptr(sv), _dptr(tv)))
        return ob.view(np.bool_), sv, tv, di

    def predict_batch(self, params, mode, ego4, k_count, other_x, other_v, selected_speed, dt, min_crash_distance, want_acc=False):
        ego4 = np.ascontiguousarray(ego4, dtype=np.float64)
        N = ego4.shape[0]
        k_count = np.ascontiguousarray(k_count, dtype=np.int32)
        other_x = np.ascontiguousarray(other_x, dtype=np.float64).reshape(N, -1)
        other_v = np.ascontiguousarray(other_v, dtype=np.float64).reshape(N, -1)
        Kmax = other_x.shape[1]
        sel = np.ascontiguousarray(selected_speed, dtype=np.float64) if selected_speed is not None else None
        eo = np.empty_like(ego4)
        xo = np.array(other_x, copy=True)
        vo = np.array(other_v, copy=True)
        cr = np.empty(N, dtype=np.int32)
        ao = np.zeros_like(other_x) if want_acc else None
        args = (self._h, C.byref(params), int(mode), N, Kmax, _dptr(ego4), _iptr(k_count), _dptr(other_x) if Kmax else None,
                _dptr(other_v) if Kmax else None, _dptr(sel), float(dt), float(min_crash_distance), _dptr(eo),
                _dptr(xo) if Kmax else None, _dptr(vo) if Kmax else None, _iptr(cr))
        if want_acc:
            self._chk(self._lib.stmpc_predict_batch_acc(*args, _dptr(ao) if Kmax else None))
        else:
            self._chk(self._lib.stmpc_predict_batch(*args))
        if want_acc:
            return eo, xo, vo, cr, ao
        return eo, xo, vo, cr

    # -- combined controller (device pointers; see include/stmpc.h) ------------------------------
    def rollout_step_device(self, params, cfg, N, Kmax, step, d_ego5_start, d_cur_ego4, d_k, d_cur_ox, d_cur_ov, d_cur_oa, d_action, stream=0):
        self._chk(self._lib.stmpc_rollout_step_device(self._h, C.byref(params), C.byref(cfg), int(N), int(Kmax), int(step), d_ego5_start, d_cur_ego4,
                                                      d_k, d_cur_ox, d_cur_ov, d_cur_oa, d_action, stream))

    def combined_decide_device(self, params, cfg, N, Kmax, d_ego5_start, d_k, d_ox_start, d_ov_start, d_cur_ego4, d_cur_ox, d_cur_ov,
                               d_first_action, d_last_choice_rl, d_takeover, d_reason, d_speed, stream=0):
        self._chk(self._lib.stmpc_combined_decide_device(self._h, C.byref(params), C.byref(cfg), int(N), int(Kmax), d_ego5_start, d_k, d_ox_start,
                                                         d_ov_start, d_cur_ego4, d_cur_ox, d_cur_ov, d_first_action, d_last_choice_rl,
                                                         d_takeover, d_reason, d_speed, stream))

    def policy_features_device(self, fcfg, N, Kmax, step, d_cur_ego4, d_k, d_cur_ox, d_cur_ov, d_cur_oa, d_evals, d_feat, feat_stride, stream=0):
        """The policy's float32 input vectors (dqn.get_state_vector_from_base_state + TimeFeature) into ``d_feat`` [N][feat_stride]."""
        self._chk(self._lib.stmpc_policy_features_device(self._h, C.byref(fcfg), int(N), int(Kmax), int(step), d_cur_ego4, d_k, d_cur_ox, d_cur_ov,
                                                         d_cur_oa, d_evals, d_feat, int(feat_stride), stream))

    def actor_create(self, w):
        """Pack and upload a policy network (dict of numpy float32 arrays w0, b0, w1, b1, w2, b2 as torch stores them + tanh_scale, tanh_mean);
        returns an opaque handle for ``actor_eval_device`` / ``actor_destroy``."""
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        w0, b0, w1, b1, w2, b2 = f32(w["w0"]), f32(w["b0"]), f32(w["w1"]), f32(w["b1"]), f32(w["w2"]).reshape(-1), f32(w["b2"]).reshape(-1)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()
        self._chk(self._lib.stmpc_actor_create(self._h, int(w0.shape[1]), int(w0.shape[0]), int(w1.shape[0]), fp(w0), fp(b0), fp(w1), fp(b1), fp(w2), fp(b2),
                                               float(w["tanh_scale"]), float(w["tanh_mean"]), C.byref(h)))
        return h

    def actor_destroy(self, handle):
        self._lib.stmpc_actor_destroy(handle)

    def actor_eval_device(self, handle, fcfg, N, Kmax, step, d_cur_ego4, d_k, d_cur_ox, d_cur_ov, d_cur_oa, d_evals, d_feat, feat_stride, d_jerk, stream=0):
        """One launch: state vectors + the packed network -> proposed jerk [N] fp64 (``stmpc_actor_eval_device``)."""
        self._chk(self._lib.stmpc_actor_eval_device(self._h, handle, C.byref(fcfg), int(N), int(Kmax), int(step), d_cur_ego4, d_k, d_cur_ox, d_cur_ov,
                                                    d_cur_oa, d_evals, d_feat, int(feat_stride), d_jerk, stream))

    def combined_counts(self, reset=False):
        """(decisions taken, controller solves run for them) since the last reset."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.stmpc_combined_counts(self._h, C.byref(a), C.byref(b), int(bool(reset))))
        return int(a.value), int(b.value)

    # -- batched episode simulator --------------------------------------------------------------
    def sim_init(self, cfg, N, stream=0):
        self._chk(self._lib.stmpc_sim_init_device(self._h, C.byref(cfg), int(N), stream))

    def sim_view(self, cfg, N, Kmax, d_ego5, d_k, d_ox, d_ov, d_oa=0, stream=0):
        self._chk(self._lib.stmpc_sim_view_device(self._h, C.byref(cfg), int(N), int(Kmax), d_ego5, d_k, d_ox, d_ov, d_oa, stream))

    def sim_step(self, params, cfg, N, d_cmd_speed, stream=0):
        self._chk(self._lib.stmpc_sim_step_device(self._h, C.byref(params), C.byref(cfg), int(N), d_cmd_speed, stream))

    def sim_status_device(self, N, d_status, stream=0):
        """Environment status words into a device int32 array (asynchronous): 0 running, 1 arrived, 2 crashed, 3 out of time."""
        self._chk(self._lib.stmpc_sim_status_device(self._h, int(N), d_status, stream))

    def check_error(self):
        """Synchronise and raise what kernels of earlier (asynchronous) calls on this context flagged; see ``stmpc_check_error``."""
        self._chk(self._lib.stmpc_check_error(self._h))

    def sim_read(self, N):
        status, ticks = np.zeros(N, np.int32), np.zeros(N, np.int32)
        acc, ego4 = np.zeros((N, SIM_NACC)), np.zeros((N, 4))
        self._chk(self._lib.stmpc_sim_read(self._h, int(N), _iptr(status), _iptr(ticks), _dptr(acc), _dptr(ego4)))
        return status, ticks, acc, ego4

    def combined_read_state(self, N, Kmax, rollout_length, after_decide=True):
        """Host copies of the rollout bookkeeping (and, after the decision, of the probe / controller results)."""
        K = max(int(Kmax), 1)
        R1 = max(int(rollout_length), 1) + 1
        out = {"live": np.zeros(N, np.int32), "hist_len": np.zeros(N, np.int32), "crash_pred": np.zeros(N, np.int32),
               "sel_speed": np.zeros(N), "rollout_s": np.zeros((N, R1)), "have_test": np.zeros(N, np.int32),
               "test_ego4": np.zeros((N, 4)), "test_ox": np.zeros((N, K)), "test_ov": np.zeros((N, K))}
        extra = {"probe_crash": np.zeros(N, np.int32), "st_speed": np.zeros(N), "fine": np.zeros((N, QP_NMAX)), "fine_len": np.zeros(N, np.int32)} if after_decide else {}
        self._chk(self._lib.stmpc_combined_read_state(self._h, int(N), _iptr(out["live"]), _iptr(out["hist_len"]), _iptr(out["crash_pred"]), _dptr(out["sel_speed"]),
                                                      _dptr(out["rollout_s"]), _iptr(out["have_test"]), _dptr(out["test_ego4"]), _dptr(out["test_ox"]), _dptr(out["test_ov"]),
                                                      _iptr(extra.get("probe_crash")), _dptr(extra.get("st_speed")), _dptr(extra.get("fine")), _iptr(extra.get("fine_len"))))
        out.update(extra)
        return out

    def probe_arith(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
        out = np.empty_like(a)
        self._chk(self._lib.stmpc_probe_arith(self._h, int(op), _dptr(a), _dptr(b), _dptr(out), a.size))
        return out


_default_ctx = None


def default_context():
    """Process-wide context on the current HIP device (created on first use)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(-1)
    return _default_ctx
