"""the reference's non-production solvers st_cy.solve_s_t_path_no_jerk_fast / _djikstra (st_cy.pyx:96-312).
Goldens = outputs of the reference's own compiled functions (tests/golden/make_golden_nojerk.py)."""
import numpy as np
import pytest

from conftest import load_golden


def _cases():
    g = load_golden("golden_nojerk.npz")
    for c in range(int(g["n_cases"])):
        yield c, g["c%d_obstacles" % c], g["c%d_s_values" % c], g["c%d_t_values" % c], float(g["c%d_v0" % c]), g["c%d_distances" % c], \
            g["c%d_fast" % c], g["c%d_djikstra" % c]


def test_oracle_matches_reference():
    from oracle import nj_oracle as nj
    n_trunc = n_differ = 0
    for c, ob, sv, tv, v0, di, fast, dj in _cases():
        assert np.array_equal(nj.solve_no_jerk("fast", ob, sv, tv, v0, di)[0], fast), c
        assert np.array_equal(nj.solve_no_jerk("djikstra", ob, sv, tv, v0, di)[0], dj), c
        n_trunc += fast[-1] == 0
        n_differ += not np.array_equal(fast, dj)
    assert n_trunc >= 5 and n_differ >= 3            # failure cases and cases where the two searches disagree are covered


@pytest.mark.gpu
def test_gpu_matches_reference(gpu_ctx, restore_settings):
    from rl_mpc_lanemerging_amd import st
    for c, ob, sv, tv, v0, di, fast, dj in _cases():
        assert np.array_equal(st.solve_s_t_path_no_jerk_fast(ob, sv, tv, v0, di), fast), c
        assert np.array_equal(st.solve_s_t_path_no_jerk_djikstra(ob, sv, tv, v0, di), dj), c


@pytest.mark.gpu
def test_gpu_dispatch_without_fast_solver(gpu_ctx, restore_settings):
    """USE_FAST_ST_SOLVER = False (st.py:749-753) on a lattice small enough for the (t, s, s_prev) search: same as the oracle on the same grids."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import st
    from rl_mpc_lanemerging_amd.prediction import HighwayState
    from oracle import nj_oracle as nj
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(dict(S_DISCRETIZATION=0.5, FUTURE_S=120.0, T_DISCRETIZATION=0.5, FUTURE_T=4.0, USE_FAST_ST_SOLVER=False))
    state = HighwayState((20.0, -1.6), 12.0, 0.0, [60.0, 35.0, -10.0], [7.0, 7.0, 7.0], [0.0, 0.0, 0.0])
    seq, ob, sv, tv, di = st.get_appropriate_base_st_path_and_obstacles(state)
    want, _ = nj.solve_no_jerk("djikstra", ob, sv, tv, state.ego_speed, di)
    assert np.array_equal(seq, want) and seq[-1] > seq[0]
    with pytest.raises(Exception):
        st.solve_s_t_path_no_jerk_djikstra(np.zeros((64, 3000), bool), np.arange(3000.0), np.arange(64.0), 1.0, np.ones((64, 3000)))
