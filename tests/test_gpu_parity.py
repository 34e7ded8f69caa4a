"""GPU parity tests: the HIP path (through the C-ABI) against the golden vectors of the
reference and against the CPU oracle on seeded inputs.  Bit-exact on every integer output
and on the fp64 cost (the north_star tolerance is 1e-6 relative; we assert equality)."""
import os

import numpy as np
import pytest

from conftest import load_golden, settings_from_golden

pytestmark = pytest.mark.gpu

STATE_FILES = ["golden_default.npz", "golden_uncertainty.npz", "golden_h40a21.npz", "golden_h40a21_unc.npz"]


def _check(res, ref, H):
    assert np.array_equal(res["best_t"], ref["best_t"])
    assert np.array_equal(res["path_idx"], ref["path_idx"])
    assert np.array_equal(res["cost"], ref["cost"])
    assert np.array_equal(res["crash"], ref["crash"])
    assert np.array_equal(np.isnan(res["path_dist"]), np.isnan(ref["path_dist"]))
    m = ~np.isnan(ref["path_dist"])
    assert np.array_equal(res["path_dist"][m], ref["path_dist"][m])


def test_device_arithmetic_is_ieee(gpu_ctx):
    """fp64 divide / sqrt / fma on the device are correctly rounded (same bits as the host)."""
    rng = np.random.default_rng(0)
    n = 1 << 20
    a = rng.uniform(-400, 400, n) * 10.0 ** rng.integers(-6, 6, n)
    b = rng.uniform(0.01, 400, n) * 10.0 ** rng.integers(-4, 4, n)
    assert np.array_equal(gpu_ctx.probe_arith(0, a, b), a / b)
    assert np.array_equal(gpu_ctx.probe_arith(1, np.abs(a)), np.sqrt(np.abs(a)))
    assert np.array_equal(gpu_ctx.probe_arith(2, a, b), a * b)
    assert np.array_equal(gpu_ctx.probe_arith(3, a, b), a + b)
    # divisions by the lattice constants: generic IEEE division and the 5-op constant division (divc<true>)
    for d in (0.3, 0.3 * 0.3, 0.3 ** 3, 0.05, 0.05000000000000071, 0.2, 0.2 ** 3, 0.1, 0.25, 0.5 ** 3):
        bb = np.full(n, d)
        assert np.array_equal(gpu_ctx.probe_arith(0, a, bb), a / bb)
        assert np.array_equal(gpu_ctx.probe_arith(5, a, bb), a / bb)
    # and by per-episode lattice steps delta_s = s_values[1] - s_values[0]
    s0 = rng.uniform(-260, 120, n)
    delta = (s0 + 0.05) - s0
    assert np.array_equal(gpu_ctx.probe_arith(5, a, delta), a / delta)


@pytest.mark.parametrize("fname", STATE_FILES)
def test_batch_matches_reference_golden(fname, gpu_ctx, restore_settings):
    from rl_mpc_lanemerging_amd import st
    g = load_golden(fname)
    p, op = settings_from_golden(g)
    res = st.solve_arrays(g["ego"], g["k_count"], g["other_x"], g["other_v"], p, gpu_ctx)
    _check(res, g, g["t_values"].size)
    # s_sequence exactly as the reference returns it
    for i in range(0, g["ego"].shape[0], 7):
        seq = st.s_sequence_from_path(res["path_idx"][i], g["ego"][i, 4], p)
        assert np.array_equal(seq, g["s_sequence"][i])


@pytest.mark.parametrize("prune,band", [("0", "0"), ("1", "0"), ("1", "3"), ("1", "100000"), ("2", "0"), ("1", "nodense")])
def test_bounded_search_is_exact(prune, band, restore_settings, monkeypatch):
    """The banded pre-pass + bounded exact pass returns the same bits as the unbounded DP, whatever the band
    (a tiny band makes the bound loose or absent, a huge one makes the pre-pass the full search)."""
    from rl_mpc_lanemerging_amd import _capi, st
    monkeypatch.setenv("STMPC_PRUNE", "1" if prune == "2" else prune)
    if prune == "2":
        monkeypatch.setenv("STMPC_TWO_PHASE", "1")      # bound-only phase, heaviest-first order, exact phase
    if band == "nodense":
        monkeypatch.setenv("STMPC_BAND_DENSE", "0")      # the general pass (dp_pass) runs the bounding attempts instead of the dense one (band_pass)
    elif band != "0":
        monkeypatch.setenv("STMPC_BAND", band)
    ctx = _capi.Context(0)
    for fname in STATE_FILES:
        g = load_golden(fname)
        p, op = settings_from_golden(g)
        res = st.solve_arrays(g["ego"], g["k_count"], g["other_x"], g["other_v"], p, ctx)
        _check(res, g, g["t_values"].size)
    ctx.close()


@pytest.mark.parametrize("last_infl,band", [("1", "450"), ("1.005", "450"), ("1.5", "450"), ("1.005", "60"), ("4", "900")])
def test_retry_bound_from_the_last_layer_is_exact(last_infl, band, restore_settings, monkeypatch):
    """A narrow bounding band makes a tenth of the bounds fall below the reference's answer, so the exact pass fails and repeats.  The repeat is bounded
    by the cheapest node of the failed pass's last layer (terminals up to STMPC_LAST_INFL x the bound are recorded; 1 = off: the growth ladder): every
    output bit is the reference's whatever the factor, on a batch large enough to hold a few hundred repeats, with every episode against the oracle."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    monkeypatch.setenv("STMPC_LAST_INFL", last_infl)
    monkeypatch.setenv("STMPC_BAND", band)
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    n = 1536
    ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=77)
    ctx = _capi.Context(0)
    res = st.solve_arrays(ego, kc, ox, ov, p, ctx)
    assert ctx.stats()["retries"] >= n // 40            # (the case under test does occur)
    ref = orc.solve_batch(orc.OrcParams.from_dict(p.as_dict()), ego, kc, ox, ov, solver="layered", nthreads=8)
    _check(res, ref, _capi.num_t(p))
    assert np.array_equal(res["cost"], ref["cost"])
    ctx.close()
    for fname in STATE_FILES:
        g = load_golden(fname)
        p2, op = settings_from_golden(g)
        ctx = _capi.Context(0)
        _check(st.solve_arrays(g["ego"], g["k_count"], g["other_x"], g["other_v"], p2, ctx), g, g["t_values"].size)
        ctx.close()


@pytest.mark.parametrize("tube,dense", [("0", "1"), ("8", "1"), ("96", "1"), ("96", "0"), ("127", "1"), ("4000", "1")])
def test_guided_bounding_attempt_is_exact(tube, dense, restore_settings, monkeypatch):
    """The guided bounding attempt (a tube around the unobstructed optimum, guide cells from the predictor kernel) only supplies a bound: whatever
    the tube's width -- off, too narrow to hold a path, the default, the widest the dense pass takes (one lane per cell: 255 cells), wider than the
    lattice -- and whichever of the two implementations runs it (tube_pass, or dp_pass under a tube), every output bit is the reference's; and on the
    wide lattice it does supply the bound for a good part of the states."""
    from rl_mpc_lanemerging_amd import _capi, st
    monkeypatch.setenv("STMPC_TUBE", tube)
    monkeypatch.setenv("STMPC_TUBE_DENSE", dense)
    ctx = _capi.Context(0)
    for fname in STATE_FILES:
        g = load_golden(fname)
        p, op = settings_from_golden(g)
        res = st.solve_arrays(g["ego"], g["k_count"], g["other_x"], g["other_v"], p, ctx)
        _check(res, g, g["t_values"].size)
        if fname == "golden_h40a21.npz":
            guided = ctx.stats()["guided"]
            assert guided == 0 if tube == "0" else guided > g["ego"].shape[0] // 8
    ctx.close()


def test_general_lattice_form_routing(restore_settings, monkeypatch):
    """Episodes whose lattice is not start + n*delta (not reachable through np.arange, but guarded) are sent to the
    last tier, which evaluates the general form: force that route for every episode and compare."""
    from rl_mpc_lanemerging_amd import _capi, st
    monkeypatch.setenv("STMPC_FORCE_GENERAL", "1")
    ctx = _capi.Context(0)
    for fname in ("golden_default.npz", "golden_h40a21.npz"):
        g = load_golden(fname)
        p, op = settings_from_golden(g)
        res = st.solve_arrays(g["ego"], g["k_count"], g["other_x"], g["other_v"], p, ctx)
        _check(res, g, g["t_values"].size)
        assert ctx.stats()["fast_path"] == 0
    ctx.close()


@pytest.mark.parametrize("tiers", ["64", "256,512", "512,2048", "64,128,256"])
@pytest.mark.parametrize("fastdiv", ["1", "0"])
def test_window_overflow_falls_back_exactly(tiers, fastdiv, restore_settings, monkeypatch):
    """Tiny LDS windows force episodes through the larger-window / HBM-scratch tiers: results must not
    change; neither may they change between the 5-op constant division and the generic IEEE division."""
    from rl_mpc_lanemerging_amd import _capi, st
    monkeypatch.setenv("STMPC_TIERS", tiers)
    monkeypatch.setenv("STMPC_FASTDIV", fastdiv)
    ctx = _capi.Context(0)
    g = load_golden("golden_default.npz")
    p, op = settings_from_golden(g)
    res = st.solve_arrays(g["ego"], g["k_count"], g["other_x"], g["other_v"], p, ctx)
    _check(res, g, g["t_values"].size)
    s = ctx.stats()
    assert s["fallback"] > 0
    if tiers == "64":
        assert s["hbm_tier"] == s["fallback"]
    ctx.close()


def test_batch_matches_oracle_seeded(gpu_ctx, restore_settings):
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    for seed, k, vary in ((101, 6, False), (102, 8, True), (103, 0, False), (104, 20, True)):
        ego, kc, ox, ov = synth.generate_states(512, k=k, kmax=max(k, 1), seed=seed, vary_k=vary, blocked_quota=0.1)
        res = st.solve_arrays(ego, kc, ox, ov, p, gpu_ctx)
        ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=8)
        _check(res, ref, _capi.num_t(p))


def test_h40a21_matches_oracle_seeded(gpu_ctx, restore_settings):
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    assert _capi.num_t(p) == 40 and _capi.num_s(p, 0.0) == 7201
    op = orc.OrcParams.from_dict(p.as_dict())
    ego, kc, ox, ov = synth.generate_states(192, k=6, kmax=8, seed=7)
    res = st.solve_arrays(ego, kc, ox, ov, p, gpu_ctx)
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=8)
    _check(res, ref, 40)


@pytest.mark.parametrize("over,k,kmax", [
    (dict(MAX_POSITIVE_ACCELERATION=5.2, MINIMUM_NEGATIVE_JERK=-35.0, MAXIMUM_POSITIVE_JERK=35.0), 6, 8),     # fan-out 21, H=18
    (dict(FUTURE_T=2.0, T_DISCRETIZATION=0.5, S_DISCRETIZATION=0.1, FUTURE_S=60.0), 12, 16),                 # H=5, S=601, K>8
    (dict(S_DISCRETIZATION=0.025, FUTURE_S=100.0, FUTURE_T=3.0), 3, 4),                                      # S=4001, fan-out ~11
    (dict(MAX_SPEED=12, DESIRED_SPEED=10.0, V_WEIGHT=0.0, A_WEIGHT=0.0, J_WEIGHT=0.0), 6, 8),                # only the gap term: many ties
])
def test_other_parameter_sets_match_oracle(over, k, kmax, gpu_ctx, restore_settings):
    """Kernel variants selected by the parameters (wide/narrow fan-out, staged/unstaged vehicle table, K > 8)."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(over)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    ego, kc, ox, ov = synth.generate_states(384, k=k, kmax=kmax, seed=77, vary_k=True, dt=p.dt)
    res = st.solve_arrays(ego, kc, ox, ov, p, gpu_ctx)
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="heap", nthreads=8)
    _check(res, ref, _capi.num_t(p))


def test_raw_grid_entry_matches_st_cy(gpu_ctx):
    from rl_mpc_lanemerging_amd import st
    g = load_golden("golden_rawgrid.npz")
    for c in range(int(g["n_cases"])):
        v0, a0 = g["c%d_v0a0" % c]
        seq = st.solve_s_t_path_fast(g["c%d_obstacles" % c], g["c%d_s_values" % c], g["c%d_t_values" % c], v0, a0,
                                     g["c%d_distances" % c], *g["c%d_tunables" % c])
        assert np.array_equal(seq, g["c%d_s_sequence" % c]), "case %d" % c


def test_grid_build_and_state_entry_match_reference(gpu_ctx, restore_settings):
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import st
    from rl_mpc_lanemerging_amd.prediction import HighwayState
    g = load_golden("golden_default.npz")
    p, op = settings_from_golden(g)
    for gi in range(int(g["n_full_grids"])):
        k = int(g["k_count"][gi])
        state = HighwayState((g["ego"][gi, 0], g["ego"][gi, 1]), g["ego"][gi, 2], g["ego"][gi, 3],
                             list(g["other_x"][gi, :k]), list(g["other_v"][gi, :k]), [0.0] * k)
        s_seq, ob, sv, tv, di = st.get_appropriate_base_st_path_and_obstacles(state)
        ref_ob = np.unpackbits(g["grid%d_obstacles_packed" % gi])[:ob.size].reshape(ob.shape).astype(bool)
        assert np.array_equal(ob, ref_ob)
        assert np.array_equal(di, g["grid%d_distances" % gi])
        assert np.array_equal(sv, g["grid%d_s_values" % gi])
        assert np.array_equal(tv, g["t_values"])
        assert np.array_equal(s_seq, g["s_sequence"][gi])
        assert st.test_guaranteed_crash_from_state(state) == bool(g["crash"][gi])
        # the materialised-grid entry agrees with the fused one
        seq2 = st.solve_s_t_path_fast(ob, sv, tv, state.ego_speed, state.ego_acceleration, di,
                                      pkg.Settings.D_WEIGHT, pkg.Settings.V_WEIGHT, pkg.Settings.A_WEIGHT,
                                      pkg.Settings.J_WEIGHT, pkg.Settings.DESIRED_SPEED, pkg.Settings.MAX_SPEED,
                                      pkg.Settings.MAX_NEGATIVE_ACCELERATION, pkg.Settings.MAX_POSITIVE_ACCELERATION,
                                      pkg.Settings.MINIMUM_NEGATIVE_JERK, pkg.Settings.MAXIMUM_POSITIVE_JERK,
                                      pkg.Settings.MIN_ALLOWED_DISTANCE)
        assert np.array_equal(seq2, g["s_sequence"][gi])


def test_predictor_steps_match_reference(gpu_ctx, restore_settings):
    g = load_golden("golden_predictor.npz")
    d = load_golden("golden_default.npz")
    p, op = settings_from_golden(d)
    for dt in (0.2, 0.3):
        for mcd in (5.0, 5.1, 3.0, 7.5):
            m = (g["dt"] == dt) & (g["mcd"] == mcd)
            if not m.any():
                continue
            eo, xo, vo, cr = gpu_ctx.predict_batch(p, 0, g["ego"][m, :4], g["k_count"][m], g["other_x"][m], g["other_v"][m],
                                                   g["sel"][m], dt, mcd)
            kk = g["k_count"][m]
            mask = np.arange(g["other_x"].shape[1])[None, :] < kk[:, None]
            assert np.array_equal(eo, g["with_ego"][m])
            assert np.array_equal(xo[mask], g["with_x"][m][mask]) and np.array_equal(vo[mask], g["with_v"][m][mask])
            assert np.array_equal(cr, g["with_crash"][m])
            eo, xo, vo, cr = gpu_ctx.predict_batch(p, 1, g["ego"][m, :4], g["k_count"][m], g["other_x"][m], g["other_v"][m],
                                                   None, dt, mcd)
            assert np.array_equal(eo, g["without_ego"][m])
            assert np.array_equal(xo[mask], g["without_x"][m][mask]) and np.array_equal(vo[mask], g["without_v"][m][mask])
            assert np.array_equal(cr, g["without_crash"][m])


def test_predictor_thresholds_match_reference(gpu_ctx, restore_settings):
    """Predicted ego within 1e-9 of the reaction (8) / crash (11) thresholds: the device's arclength map (x*x for the
    squares where CPython calls pow) must take the reference's side of every comparison."""
    g = load_golden("golden_thresholds.npz")
    d = load_golden("golden_default.npz")
    p, op = settings_from_golden(d)
    for dt in (0.2, 0.3):
        m = g["dt"] == dt
        eo, xo, vo, cr = gpu_ctx.predict_batch(p, 0, g["ego"][m, :4], g["k_count"][m], g["other_x"][m], g["other_v"][m], g["sel"][m], dt, 5.0)
        kk = g["k_count"][m]
        mask = np.arange(g["other_x"].shape[1])[None, :] < kk[:, None]
        assert np.array_equal(eo, g["with_ego"][m])
        assert np.array_equal(xo[mask], g["with_x"][m][mask]) and np.array_equal(vo[mask], g["with_v"][m][mask])
        assert np.array_equal(cr, g["with_crash"][m])


def test_config4_shard_properties(gpu_ctx, restore_settings):
    """BASELINE configs[3] per-rank shard: 8192 episodes (65536 over 8 GPUs) at H=40, S=7201 on one GPU --
    idempotence, sub-batch consistency, structure of every path, and 128 episodes bit-for-bit against the heap restatement."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    H = _capi.num_t(p)
    n = 8192
    ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=1003)
    r1 = st.solve_arrays(ego, kc, ox, ov, p, gpu_ctx)
    r2 = st.solve_arrays(ego, kc, ox, ov, p, gpu_ctx)
    for key in ("path_idx", "best_t", "cost", "crash"):
        assert np.array_equal(r1[key], r2[key])
    r4 = st.solve_arrays(ego[4096:4096 + 257], kc[4096:4096 + 257], ox[4096:4096 + 257], ov[4096:4096 + 257], p, gpu_ctx)
    assert np.array_equal(r4["path_idx"], r1["path_idx"][4096:4096 + 257]) and np.array_equal(r4["cost"], r1["cost"][4096:4096 + 257])
    path, bt = r1["path_idx"], r1["best_t"]
    valid = np.arange(H)[None, :] <= bt[:, None]
    assert (path[:, 0] == 0).all() and ((path >= 0) == valid).all()
    dd = np.diff(path, axis=1)
    assert (dd[valid[:, 1:]] >= 0).all() and (dd[valid[:, 1:]] <= p.v_max * p.dt / p.ds + 1).all()
    assert ((bt < H - 1) <= (r1["crash"] == 1)).all()
    op = orc.OrcParams.from_dict(p.as_dict())
    sel = np.arange(0, n, 64)
    ref = orc.solve_batch(op, ego[sel], kc[sel], ox[sel], ov[sel], solver="heap", nthreads=8)
    assert np.array_equal(ref["path_idx"], path[sel]) and np.array_equal(ref["cost"], r1["cost"][sel]) and np.array_equal(ref["best_t"], bt[sel])


def test_bench_forced_rccl_single_rank():
    """bench.py's collective path on one GPU: a 1-rank RCCL group, all_gather of (action, cost) checked against the solver's outputs."""
    import json, subprocess, sys
    env = dict(os.environ, STMPC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"),
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--episodes", "512"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and "all_gather" in line["config"]["collective"] and line["value"] > 0
    assert line["rccl_ranks"] == 1                     # the rank count RCCL itself reports
    assert line["config4_shard"]["episodes_per_gpu"] == 8192 and line["config4_shard"]["value"] > 0      # BASELINE configs[3]'s per-rank shard, timed in every multi-rank run
    assert line["host"]["host_us_per_step"] > 0 and line["host"]["cpus_in_affinity_mask"] >= 1           # what a rank's host side costs (the 1 -> 8 GPU curve's other risk)


def test_edge_cases(gpu_ctx, restore_settings):
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    # empty batch
    r = st.solve_arrays(np.zeros((0, 5)), np.zeros(0, np.int32), np.zeros((0, 1)), np.zeros((0, 1)), p, gpu_ctx)
    assert r["path_idx"].shape == (0, _capi.num_t(p))
    # no vehicles at all, standing start, ego far up the ramp / far down the highway
    ego = np.array([[-250.0, 28.4, 0.0, 0.0, 0.0], [59.0, -1.6, 25.0, 4.5, 0.0], [1.49, -1.59, 30.0, -6.0, 0.0]])
    for i in range(3):
        ego[i, 4] = _capi.ego_s(ego[i, 0], ego[i, 1])
    kc = np.zeros(3, np.int32)
    r = st.solve_arrays(ego, kc, np.zeros((3, 1)), np.zeros((3, 1)), p, gpu_ctx)
    ref = orc.solve_batch(op, ego, kc, np.zeros((3, 1)), np.zeros((3, 1)), solver="heap")
    _check(r, ref, _capi.num_t(p))
    # k_count out of range is rejected
    with pytest.raises(_capi.StmpcError):
        st.solve_arrays(ego, np.array([2, 0, 0], np.int32), np.zeros((3, 1)), np.zeros((3, 1)), p, gpu_ctx)


def test_full_size_batch_properties(gpu_ctx, restore_settings):
    """BASELINE-size batch (4096 episodes, H=40, S=7201, fan-out 21): size-independent properties, then ALL 4096 episodes bit for bit against
    the oracle's layered DP (paths, deepest layer, cost bits, crash verdict) and every 16th against its literal heap Dijkstra."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    H = _capi.num_t(p)
    n = 4096
    ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=4242)
    r1 = st.solve_arrays(ego, kc, ox, ov, p, gpu_ctx)
    # idempotence
    r2 = st.solve_arrays(ego, kc, ox, ov, p, gpu_ctx)
    for key in ("path_idx", "best_t", "cost", "crash"):
        assert np.array_equal(r1[key], r2[key])
    # permutation of the episodes permutes the results (no cross-episode coupling, any work order)
    perm = np.random.default_rng(1).permutation(n)
    r3 = st.solve_arrays(ego[perm], kc[perm], ox[perm], ov[perm], p, gpu_ctx)
    for key in ("path_idx", "best_t", "cost", "crash"):
        assert np.array_equal(r1[key][perm], r3[key])
    # a sub-batch gives the same answers as the full batch
    r4 = st.solve_arrays(ego[:129], kc[:129], ox[:129], ov[:129], p, gpu_ctx)
    assert np.array_equal(r4["path_idx"], r1["path_idx"][:129]) and np.array_equal(r4["cost"], r1["cost"][:129])
    # structural properties of every returned path
    path, bt = r1["path_idx"], r1["best_t"]
    assert (path[:, 0] == 0).all()
    t_idx = np.arange(H)[None, :]
    valid = t_idx <= bt[:, None]
    assert ((path >= 0) == valid).all()
    d = np.diff(path, axis=1)
    assert (d[valid[:, 1:]] >= 0).all()                        # speeds are >= 0
    vmax_cells = p.v_max * p.dt / p.ds + 1
    assert (d[valid[:, 1:]] <= vmax_cells).all()               # speed limit
    assert np.isfinite(r1["cost"]).all() and (r1["cost"][bt > 0] > 0).all()
    assert ((bt < H - 1) <= (r1["crash"] == 1)).all()          # a truncated path is always reported as a guaranteed crash
    # every episode against the oracle (layered DP, ~2 s on the box's 16 usable cores); every 16th against the literal heap restatement
    op = orc.OrcParams.from_dict(p.as_dict())
    _assert_batch_equals_oracle(orc, op, ego, kc, ox, ov, r1, heap_every=16)


def _assert_batch_equals_oracle(orc, op, ego, kc, ox, ov, res, heap_every):
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=16)
    for key in ("path_idx", "best_t", "crash"):
        assert np.array_equal(ref[key], res[key]), key
    assert np.array_equal(ref["cost"].view(np.uint64), res["cost"].view(np.uint64)), "cost bits"
    sel = np.arange(0, ego.shape[0], heap_every)
    heap = orc.solve_batch(op, ego[sel], kc[sel], ox[sel], ov[sel], solver="heap", nthreads=16)
    assert np.array_equal(heap["path_idx"], res["path_idx"][sel]) and np.array_equal(heap["best_t"], res["best_t"][sel])
    assert np.array_equal(heap["cost"].view(np.uint64), res["cost"][sel].view(np.uint64)) and np.array_equal(heap["crash"], res["crash"][sel])


def test_fused_action_cost_rows_and_one_8192_shard(gpu_ctx, restore_settings):
    """BASELINE configs[3]'s per-rank shard (8192 of 65536 episodes, H=40): every episode against the oracle, every 16th against the heap; and
    the fused (action, cost) rows the solver writes for the multi-GPU gather (stmpc_solve_batch_device_ac) equal (path_idx[:, 1], cost)."""
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    H = _capi.num_t(p)
    n = 8192
    ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=2000)            # rank 0's shard of bench.py's config4 leg
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.as_tensor(a, device=dev)
    d_ego, d_k, d_ox, d_ov = t(ego), t(kc), t(ox), t(ov)
    d_path = torch.empty((n, H), dtype=torch.int32, device=dev); d_bt = torch.empty(n, dtype=torch.int32, device=dev)
    d_cost = torch.empty(n, dtype=torch.float64, device=dev); d_crash = torch.empty(n, dtype=torch.int32, device=dev)
    d_ac = torch.full((n, 2), float("nan"), dtype=torch.float64, device=dev)
    gpu_ctx.solve_batch_device(p, n, 8, d_ego.data_ptr(), d_k.data_ptr(), d_ox.data_ptr(), d_ov.data_ptr(), d_path.data_ptr(), d_bt.data_ptr(),
                               d_cost.data_ptr(), 0, d_crash.data_ptr(), torch.cuda.current_stream().cuda_stream, d_ac.data_ptr())
    torch.cuda.synchronize()
    stats = gpu_ctx.stats()
    assert stats["resume_refused"] in (0, 1)
    assert torch.equal(d_ac[:, 0].to(torch.int32), d_path[:, 1]) and torch.equal(d_ac[:, 1], d_cost)
    assert bool((d_ac[:, 0] == d_ac[:, 0].round()).all())
    res = {"path_idx": d_path.cpu().numpy(), "best_t": d_bt.cpu().numpy(), "cost": d_cost.cpu().numpy(), "crash": d_crash.cpu().numpy()}
    op = orc.OrcParams.from_dict(p.as_dict())
    _assert_batch_equals_oracle(orc, op, ego, kc, ox, ov, res, heap_every=16)


def _random_overrides(rng):
    ds = float(rng.choice([0.025, 0.05, 0.1, 0.2]))
    dt = float(rng.choice([0.2, 0.25, 0.3, 0.4, 0.5]))
    H = int(rng.integers(3, 41))
    over = dict(S_DISCRETIZATION=ds, T_DISCRETIZATION=dt, FUTURE_T=round((H - 1) * dt, 6),
                FUTURE_S=float(rng.choice([40.0, 80.0, 150.0, 300.0])) * (1.0 if ds >= 0.05 else 0.5),
                V_WEIGHT=float(rng.choice([0.0, 0.5, 3.0, 10.0])), A_WEIGHT=float(rng.choice([0.0, 1.0, 10.0])),
                J_WEIGHT=float(rng.choice([0.0, 1.0, 10.0])), D_WEIGHT=float(rng.choice([0.0, 1.0, 10.0])),
                DESIRED_SPEED=float(rng.uniform(5.0, 25.0)), MAX_SPEED=float(rng.uniform(20.0, 35.0)),
                MAX_POSITIVE_ACCELERATION=float(rng.uniform(1.0, 6.0)), MAX_NEGATIVE_ACCELERATION=-float(rng.uniform(2.0, 8.0)),
                MAXIMUM_POSITIVE_JERK=float(rng.choice([2.0, 10.0, 35.0])), MINIMUM_NEGATIVE_JERK=-float(rng.choice([2.0, 10.0, 35.0])),
                MIN_ALLOWED_DISTANCE=float(rng.choice([0.0, 2.0, 5.0, 8.0])), CRASH_MIN_S=float(rng.uniform(6.0, 25.0)),
                START_UNCERTAINTY=float(rng.choice([0.0, 0.0, 0.5])), UNCERTAINTY_PER_SECOND=float(rng.choice([0.0, 0.0, 0.3])),
                MAX_PREDICTED_DECELERATION=-float(rng.uniform(1.0, 6.0)))
    return over


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("STMPC_FUZZ_SEEDS", "16")))))
def test_random_parameter_sets_match_oracle(seed, gpu_ctx, restore_settings):
    """Seeded random Settings (lattice spacing, horizon, weights incl. zeros, binding and non-binding limits,
    uncertainty growth): every kernel variant and window tier the parameters select must reproduce the oracle."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    rng = np.random.default_rng(9000 + seed)
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(_random_overrides(rng))
    p = _capi.Params.from_settings(pkg.Settings)
    H, S = _capi.num_t(p), _capi.num_s(p, 0.0)
    assert 2 <= H <= 64 and 2 <= S <= 65000
    op = orc.OrcParams.from_dict(p.as_dict())
    kmax = int(rng.choice([4, 8, 12]))
    ego, kc, ox, ov = synth.generate_states(160, k=int(rng.integers(0, kmax + 1)), kmax=kmax, seed=500 + seed, vary_k=True, dt=p.dt)
    res = st.solve_arrays(ego, kc, ox, ov, p, gpu_ctx)
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=8)
    _check(res, ref, H)


@pytest.mark.parametrize("overlap,resume,pool", [("1", "1", ""), ("0", "1", ""), ("1", "0", ""), ("0", "0", ""), ("1", "1", "7"), ("0", "1", "1")])
def test_concurrent_overflow_launch_is_exact(overlap, resume, pool, restore_settings, monkeypatch):
    """More episodes than persistent workgroups on the wide lattice: with STMPC_OVERLAP=1 the second LDS window's launch
    runs on a side stream and consumes the overflow queue while the first launch is still filling it; with
    STMPC_RESUME=1 it continues exact passes from the layer the first window checkpointed (saved, with the back-pointer rows written so
    far, in an entry of the checkpoint pool) instead of starting over; with a pool of 7 / 1 entries (STMPC_POOL) nearly all overflowing
    searches find it exhausted and start over.  Results must not depend on any of it."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    monkeypatch.setenv("STMPC_OVERLAP", overlap)
    monkeypatch.setenv("STMPC_RESUME", resume)
    if pool:
        monkeypatch.setenv("STMPC_POOL", pool)
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    ego, kc, ox, ov = synth.generate_states(1400, k=6, kmax=8, seed=4321)
    ctx = _capi.Context(0)
    for rep in range(2):                       # second call: queues and counters are re-initialised per launch
        res = st.solve_arrays(ego, kc, ox, ov, p, ctx)
    s = ctx.stats()
    assert s["fallback"] > 20 and s["hbm_tier"] == 0
    if pool:
        assert s["pool_exhausted"] >= 1 and s["pool_exhausted"] <= s["fallback"]
    else:
        assert s["pool_exhausted"] == 0
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=16)
    _check(res, ref, 40)
    ctx.close()


@pytest.mark.parametrize("name,over,n,k,kmax", [
    ("H=64", dict(T_DISCRETIZATION=0.1, FUTURE_T=6.3), 256, 6, 8),                                   # STMPC_H_LIMIT
    ("S=60001", dict(S_DISCRETIZATION=0.005, FUTURE_S=300.0, FUTURE_T=3.0), 48, 4, 4),               # near STMPC_S_LIMIT
    ("K=32", dict(), 256, 32, 32),                                                                   # STMPC_KMAX_LIMIT
    ("S=30001,H=40", dict(S_DISCRETIZATION=0.0125, FUTURE_S=375.0, FUTURE_T=7.8, T_DISCRETIZATION=0.2), 24, 6, 8),   # HBM-scratch tier in use
    ("H=2", dict(FUTURE_T=0.3), 128, 6, 8),
    ("S=4", dict(FUTURE_S=0.1), 64, 6, 8),
])
def test_limits_of_the_interface(name, over, n, k, kmax, restore_settings):
    """The documented limits (64 layers, 65000 cells, 32 vehicles) and the degenerate small ends, against the oracle."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(over)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    ego, kc, ox, ov = synth.generate_states(n, k=k, kmax=kmax, seed=len(name), vary_k=True, dt=p.dt)
    ctx = _capi.Context(0)
    res = st.solve_arrays(ego, kc, ox, ov, p, ctx)
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=16)
    _check(res, ref, _capi.num_t(p))
    if name == "S=30001,H=40":
        assert ctx.stats()["hbm_tier"] > 0
    ctx.close()


def test_batch_sizes_around_the_scheduling_thresholds(restore_settings):
    """One context, batch sizes on both sides of the thresholds that switch on task splitting, the side launch and the
    checkpointing (multiples of the persistent grid), in mixed order: buffers are re-sized, queues and counters re-set."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    ego, kc, ox, ov = synth.generate_states(2600, k=6, kmax=8, seed=99)
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=16)
    ctx = _capi.Context(0)
    for n in (2600, 1, 1025, 2047, 2049, 300, 2600, 1024):
        res = st.solve_arrays(ego[:n], kc[:n], ox[:n], ov[:n], p, ctx)
        _check(res, {q: ref[q][:n] for q in ("path_idx", "best_t", "cost", "crash", "path_dist")}, 40)
    ctx.close()


@pytest.mark.parametrize("cap", ["0", "20", "300", "100000"])
def test_band_cap_never_changes_results(cap, restore_settings, monkeypatch):
    """The beam-like control of the pre-pass band (STMPC_BAND_CAP) only moves work between the bounding and the exact
    pass: off, starved, default and never-binding caps give the same bits."""
    from rl_mpc_lanemerging_amd import _capi, st
    monkeypatch.setenv("STMPC_PRUNE", "1")
    monkeypatch.setenv("STMPC_BAND_CAP", cap)
    ctx = _capi.Context(0)
    for fname in STATE_FILES:
        g = load_golden(fname)
        p, op = settings_from_golden(g)
        res = st.solve_arrays(g["ego"], g["k_count"], g["other_x"], g["other_v"], p, ctx)
        _check(res, g, g["t_values"].size)
    ctx.close()


@pytest.mark.parametrize("gsh,heavy_first", [("0", "0"), ("1", "0"), ("3", "1"), ("4", "1")])
def test_lane_mapping_and_task_order_never_change_results(gsh, heavy_first, restore_settings, monkeypatch):
    """Lanes per source of sparse layers (STMPC_GSH: 1, 2, 8 or 16 lanes) and the static task order (STMPC_HEAVY_FIRST) only
    change who evaluates a candidate and when: every golden file and a 2300-episode wide-fan batch give the same bits."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    monkeypatch.setenv("STMPC_GSH", gsh)
    monkeypatch.setenv("STMPC_HEAVY_FIRST", heavy_first)
    ctx = _capi.Context(0)
    for fname in STATE_FILES:
        g = load_golden(fname)
        p, op = settings_from_golden(g)
        res = st.solve_arrays(g["ego"], g["k_count"], g["other_x"], g["other_v"], p, ctx)
        _check(res, g, g["t_values"].size)
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    p = _capi.Params.from_settings(pkg.Settings)
    ego, kc, ox, ov = synth.generate_states(2300, k=6, kmax=8, seed=77)
    res = st.solve_arrays(ego, kc, ox, ov, p, ctx)
    sel = np.arange(0, 2300, 23)
    ref = orc.solve_batch(orc.OrcParams.from_dict(p.as_dict()), ego[sel], kc[sel], ox[sel], ov[sel], solver="layered", nthreads=16)
    for key in ("path_idx", "best_t", "cost", "crash"):
        assert np.array_equal(res[key][sel], ref[key]), key
    ctx.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("STMPC_FUZZ_BIG_SEEDS", "4")))))
def test_random_parameter_sets_large_batches(seed, restore_settings):
    """Random wide-fan parameter sets at batch sizes where task splitting, the side launch and checkpoint/resume are all
    active (more than two tasks per persistent workgroup)."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    rng = np.random.default_rng(7000 + seed)
    over = _random_overrides(rng)
    # force a wide fan-out (bounded search on) and a lattice big enough to overflow the first window now and then
    over.update(S_DISCRETIZATION=0.05, T_DISCRETIZATION=0.3, MAX_POSITIVE_ACCELERATION=float(rng.uniform(4.0, 6.0)),
                MAXIMUM_POSITIVE_JERK=35.0, MINIMUM_NEGATIVE_JERK=-35.0, FUTURE_S=float(rng.choice([200.0, 360.0])))
    H = int(rng.integers(12, 41))
    over["FUTURE_T"] = round((H - 1) * 0.3, 6)
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(over)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    n = int(rng.integers(2100, 2600))
    ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=800 + seed, vary_k=True, dt=p.dt)
    ctx = _capi.Context(0)
    res = st.solve_arrays(ego, kc, ox, ov, p, ctx)
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=32)
    _check(res, ref, _capi.num_t(p))
    ctx.close()


@pytest.mark.parametrize("kmax", [8, 12, 20])
def test_horizon_prediction_matches_oracle_on_merge_zone_states(kmax, gpu_ctx, restore_settings):
    """k_predict's recurrence (one lane per vehicle, leader chains settled by repeated sweeps, squared-distance thresholds, the ego's
    curved step only where it moves) against the oracle's predictor applied H-1 times (prediction.py:22-105 -> st.py:25-70): the
    materialised grids of 150 states per width that put the phantom ego through every branch -- on the ramp before and after the
    reaction threshold, beside and between vehicles, ahead of all of them, stopped leaders (long deceleration chains), vehicles that
    overtake the ego's position, states at the ramp's end point and at the merge point -- bit for bit, for 8-, 16- and 32-lane groups."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    rng = np.random.default_rng(900 + kmax)
    checked = curved = 0
    for i in range(150):
        mode = i % 6
        ex = [rng.uniform(-75, -40), rng.uniform(-51.2, -50.6), rng.uniform(-45, 1.4), rng.uniform(1.3, 1.7), rng.uniform(1.5, 40), rng.uniform(-250, 60)][mode]
        ey = (1.72 + 0.134 * (-50.9 - ex)) if ex < -50.9 else ((1.71 + (ex + 50.9) / 52.4 * (-1.6 - 1.71)) if ex < 1.5 else -1.6)
        ey += rng.uniform(-0.05, 0.05) if i % 4 == 0 else 0.0
        k = int(rng.integers(0, kmax + 1)) if i % 7 else kmax
        lead = ex + rng.uniform(-30, 70)
        gaps = rng.uniform(4.0, 28.0, size=max(k, 1))
        xs = (lead - np.concatenate([[0.0], np.cumsum(gaps[:-1])]))[:k]
        vs = rng.choice([7.0, 11.0, 15.0], size=1) + rng.choice([0.0, 0.0, -1.0, -4.0, 2.5], size=k)
        if i % 5 == 0 and k:
            vs[int(rng.integers(0, k))] = 0.0                              # a stopped vehicle: everyone behind it brakes in turn
        vs = np.maximum(vs, 0.0)
        ev, ea = rng.uniform(0, 25), rng.uniform(-3, 3)
        start_s = _capi.ego_s(ex, ey)
        ob, sv, tv, di = gpu_ctx.build_grid(p, [ex, ey, ev, ea, start_s], xs, vs)
        st_ = orc.make_state(ex, ey, ev, ea, list(xs), list(vs))
        rob, rsv, rtv, rdi = orc.build_grid(op, st_, start_s)
        assert np.array_equal(ob, rob), (i, mode)
        assert np.array_equal(di, rdi), (i, mode)
        assert np.array_equal(sv, rsv)
        checked += 1
        curved += int(-45 < ex < 1.5)
    assert checked == 150 and curved > 20


def test_narrow_lattice_side_launch_follows_the_previous_batch(restore_settings, monkeypatch):
    """The reference's own lattice: the second window runs alongside the first (small side grid) only when the previous batch on the context had
    episodes for it.  With every episode routed to the last tier (STMPC_FORCE_GENERAL: the lattice form a first-window kernel is not compiled for)
    the first call finds no side launch, the later ones do; a context without the routing never starts one.  Every call gives the oracle's bits."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, st, synth
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    p = _capi.Params.from_settings(pkg.Settings)
    op = orc.OrcParams.from_dict(p.as_dict())
    n = 1400                                                                        # more episodes than persistent workgroups
    ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=1000)
    ref = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=16)
    ref = {k_: v_ for k_, v_ in ref.items() if isinstance(v_, np.ndarray)}
    plain = _capi.Context(0)
    for call in range(2):
        _check(st.solve_arrays(ego, kc, ox, ov, p, plain), ref, _capi.num_t(p))
    assert plain.stats()["fallback"] <= 8                                           # (dense layers: hardly anything overflows on this lattice)
    plain.close()
    monkeypatch.setenv("STMPC_FORCE_GENERAL", "1")
    ctx = _capi.Context(0)
    for call in range(3):
        _check(st.solve_arrays(ego, kc, ox, ov, p, ctx), ref, _capi.num_t(p))
        assert ctx.stats()["fallback"] == n
    half = st.solve_arrays(ego[:700], kc[:700], ox[:700], ov[:700], p, ctx)         # another batch size in between
    _check(half, {k_: v_[:700] for k_, v_ in ref.items()}, _capi.num_t(p))
    _check(st.solve_arrays(ego, kc, ox, ov, p, ctx), ref, _capi.num_t(p))
    ctx.close()
