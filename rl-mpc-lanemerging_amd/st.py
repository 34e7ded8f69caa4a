"""Host-side mirror of the reference's ST ("MPC") interface, backed by the HIP solver.

Same names, argument meaning and failure conventions as the reference's ``st.py`` /
``st_cy.pyx`` for the hot path, so a caller written against the reference
(``control.run_episode``'s ``control_function(state)`` protocol at control.py:310, or
``dqn.RLAgent.do_combined_control`` at dqn.py:117-200) can switch modules:

    solve_s_t_path_fast(...)                      <- st_cy.solve_s_t_path_fast          st_cy.pyx:315
    find_s_t_obstacles_from_state(...)            <- st.find_s_t_obstacles_from_state   st.py:25
    get_appropriate_base_st_path_and_obstacles(s) <- same name                          st.py:726
    do_st_control(state)                          <- same name                          st.py:757
    finer_fit(...)                                <- same name                          st.py:584
    test_guaranteed_crash_from_state(state)       <- same name                          st.py:790
    get_path_mean_abs_jerk(...)                   <- same name                          st.py:274
    get_range_index(...)                          <- same name                          st.py:20

plus the batched entry the reference does not have: ``solve_states`` / ``solve_arrays``.
All lattice work runs on the GPU through ``libstmpc.so``; nothing here falls back to a
CPU implementation.
"""
import numpy as np

from . import _capi
from . import control
from .config import Settings
from .prediction import HighwayState, pack_states  # noqa: F401  (re-export)


def get_range_index(min_s, delta_s, s):
    # st.py:20-22
    return int((s - min_s) / delta_s)


def _params_from_settings():
    return _capi.Params.from_settings(Settings)


def s_values_for(start_s, params, num_s=None):
    """The solver's s lookup table, ``np.arange(start_s, start_s + future_s + ds, ds)`` (st.py:31)."""
    if num_s is None:
        num_s = _capi.num_s(params, start_s)
    return np.arange(start_s, start_s + params.future_s + params.ds, params.ds)[:num_s]


def solve_s_t_path_fast(obstacles_bool, s_values, t_indices, ego_start_speed, ego_start_acceleration, distances,
                        d_weight, v_weight, a_weight, j_weight, desired_speed, max_speed,
                        negative_acceleration_limit, positive_acceleration_limit, negative_jerk_limit,
                        positive_jerk_limit, min_allowed_distance):
    """Drop-in for ``st_cy.solve_s_t_path_fast`` (st_cy.pyx:315-399): 17 positional arguments,
    returns a fresh ``ndarray[num_t]`` of s along the path with 0.0 past the deepest layer reached."""
    s_values = np.asarray(s_values, dtype=np.float64)
    t_indices = np.asarray(t_indices, dtype=np.float64)
    if s_values.shape[0] < 2 or t_indices.shape[0] < 2:
        raise IndexError("s_values and t_indices need at least two entries")   # st_cy.pyx:318-319
    if t_indices[1] - t_indices[0] == 0:
        raise ZeroDivisionError("float division")                             # cdivision off in the reference
    return _capi.default_context().solve_grid(
        np.asarray(obstacles_bool), s_values, t_indices, ego_start_speed, ego_start_acceleration,
        np.asarray(distances, dtype=np.float64),
        (d_weight, v_weight, a_weight, j_weight, desired_speed, max_speed, negative_acceleration_limit,
         positive_acceleration_limit, negative_jerk_limit, positive_jerk_limit, min_allowed_distance))


def find_s_t_obstacles_from_state(current_state, future_s=150, delta_s=0.5, delta_t=0.2, time_limit=5,
                                  start_uncertainty=0.0, uncertainty_per_second=0.1):
    """st.py:25-70 on the GPU: returns ``(obstacles, s_values, t_values, ego_speed, distances)``."""
    params = _params_from_settings()
    params.future_s, params.ds, params.dt, params.future_t = future_s, delta_s, delta_t, time_limit
    params.start_unc, params.unc_per_s = start_uncertainty, uncertainty_per_second
    start_s = control.get_ego_s(current_state.ego_position)
    state5 = np.array([current_state.ego_position[0], current_state.ego_position[1], current_state.ego_speed,
                       current_state.ego_acceleration, start_s], dtype=np.float64)
    obstacles, s_values, t_values, distances = _capi.default_context().build_grid(
        params, state5, np.asarray(current_state.other_xs, dtype=np.float64),
        np.asarray(current_state.other_speeds, dtype=np.float64))
    return obstacles, s_values, t_values, current_state.ego_speed, distances


def solve_arrays(ego, k_count, other_x, other_v, params=None, ctx=None, want_dist=True):
    """Batched solve on SoA arrays (see ``pack_states``).  Returns a dict with ``path_idx[N,H]``,
    ``best_t[N]``, ``cost[N]``, ``path_dist[N,H]``, ``crash[N]``."""
    params = params or _params_from_settings()
    ctx = ctx or _capi.default_context()
    path, best_t, cost, pdist, crash = ctx.solve_batch(params, ego, k_count, other_x, other_v, want_dist)
    return {"path_idx": path, "best_t": best_t, "cost": cost, "path_dist": pdist, "crash": crash}


def s_sequence_from_path(path_idx_row, start_s, params):
    """Rebuild the reference's return value: s along the path, 0.0 past ``best_t`` (st_cy.pyx:393-398)."""
    sv = s_values_for(start_s, params)
    out = np.zeros(path_idx_row.shape[0], dtype=np.float64)
    ok = path_idx_row >= 0
    out[ok] = sv[path_idx_row[ok]]
    return out


def solve_states(states, params=None, ctx=None):
    """Batched ``get_appropriate_base_st_path_and_obstacles`` without the grids:
    returns ``(s_sequences[N,H], result_dict)``."""
    params = params or _params_from_settings()
    ego, k_count, ox, ov = pack_states(states)
    res = solve_arrays(ego, k_count, ox, ov, params, ctx)
    seqs = np.stack([s_sequence_from_path(res["path_idx"][i], ego[i, 4], params) for i in range(len(states))]) \
        if len(states) else np.zeros((0, _capi.num_t(params)))
    return seqs, res


def get_appropriate_base_st_path_and_obstacles(state):
    """st.py:726-754: ``(s_sequence, obstacles, s_values, t_values, distances)`` for one state."""
    obstacles, s_values, t_values, ego_speed, distances = find_s_t_obstacles_from_state(
        state, Settings.FUTURE_S, Settings.S_DISCRETIZATION, Settings.T_DISCRETIZATION, Settings.FUTURE_T,
        Settings.START_UNCERTAINTY, Settings.UNCERTAINTY_PER_SECOND)
    if not Settings.USE_FAST_ST_SOLVER:
        # st.py:749-753: the (t, s, s_prev) search on the materialised grids, with the constants compiled into st_cy
        s_sequence = solve_s_t_path_no_jerk_djikstra(obstacles, s_values, t_values, ego_speed, distances)
        return s_sequence, obstacles, s_values, t_values, distances
    seqs, _ = solve_states([state])
    return seqs[0], obstacles, s_values, t_values, distances


def solve_s_t_path_no_jerk_fast(obstacles, s_values, t_values, ego_start_speed, distances):
    """st_cy.solve_s_t_path_no_jerk_fast (st_cy.pyx:209-312) on materialised grids: returns the s sequence, 0.0 past the deepest layer."""
    return _capi.default_context().solve_grid_no_jerk(0, obstacles, s_values, t_values, ego_start_speed, distances)


def solve_s_t_path_no_jerk_djikstra(obstacles, s_values, t_values, ego_start_speed, distances):
    """st_cy.solve_s_t_path_no_jerk_djikstra (st_cy.pyx:96-206): the search over (t, s, s_prev) nodes.  It keeps num_t * num_s**2
    node flags (the reference allocates that as well); lattices beyond 2**28 of them are rejected."""
    return _capi.default_context().solve_grid_no_jerk(1, obstacles, s_values, t_values, ego_start_speed, distances)


def finer_fit(s_sequence, delta_t, coarse_delta_t, start_speed, start_acceleration, before_after_cars=None):
    """st.py:584-723: QP re-sampling of a coarse path to the simulator tick (one wavefront per QP on the GPU).

    The reference solves the QP with ``cvxopt.solvers.qp`` (maxiters = 10, st.py:16-17); the kernel runs cvxopt's
    published coneqp iteration with the same cap and tolerances (parity of the solve is unpinned: cvxopt is not
    available to compare against).  At most 64 fine samples."""
    s_sequence = np.ascontiguousarray(s_sequence, dtype=np.float64)
    if len(s_sequence) == 1:                       # st.py:587-588
        return s_sequence
    params = _params_from_settings()
    bac = None if before_after_cars is None else np.array([before_after_cars], dtype=np.float64)
    out, out_len, _ = _capi.default_context().finer_fit_batch(
        params, delta_t, coarse_delta_t, s_sequence[None, :], [len(s_sequence)], [start_speed], [start_acceleration], bac)
    if out_len[0] < 0:
        raise ValueError("finer_fit: more than %d fine samples are not supported" % _capi.QP_NMAX)
    return out[0, :out_len[0]].copy()


def do_st_control(state):
    """st.py:757-783.  Returns the commanded speed and forwards it to ``control.set_ego_speed``."""
    ego, k_count, ox, ov = pack_states([state])
    res = _capi.default_context().st_control_batch(_params_from_settings(), Settings.TICK_LENGTH, ego, k_count, ox, ov, want_paths=True)
    if res["fine_len"][0] < 0:
        raise ValueError("finer_fit: more than %d fine samples are not supported" % _capi.QP_NMAX)
    if res["best_t"][0] != _capi.num_t(_params_from_settings()) - 1:
        print("ST Solver finds crash inevitable")              # st.py:765-766
    speed = float(res["speed"][0])
    control.set_ego_speed(speed)
    return speed


def do_st_control_batch(states, params=None, ctx=None):
    """``do_st_control`` for many independent states in one launch: returns ``(speeds[N], best_t[N])``.
    No side effect on ``control`` (there is one ego per state; the caller forwards the speeds)."""
    params = params or _params_from_settings()
    ctx = ctx or _capi.default_context()
    ego, k_count, ox, ov = pack_states(states)
    res = ctx.st_control_batch(params, Settings.TICK_LENGTH, ego, k_count, ox, ov, want_paths=True)
    if (res["fine_len"] < 0).any():
        raise ValueError("finer_fit: more than %d fine samples are not supported" % _capi.QP_NMAX)
    return res["speed"], res["best_t"]


def test_guaranteed_crash_from_state(state):
    """st.py:790-802."""
    _, res = solve_states([state])
    return bool(res["crash"][0])


test_guaranteed_crash_from_state.__test__ = False   # not a pytest test


def get_path_mean_abs_jerk(s_sequence, ego_start_speed, ego_start_acceleration, delta_t):
    """st.py:274-288: mean |jerk| of a path given the start speed and acceleration (the library's host helper, same operations)."""
    seq = np.ascontiguousarray(s_sequence, dtype=np.float64)
    if seq.size < 2:
        raise ZeroDivisionError("division by zero")        # the reference divides by len - 1
    return _capi.path_mean_abs_jerk(seq, ego_start_speed, ego_start_acceleration, delta_t)
