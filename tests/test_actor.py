"""The reference's pretrained DDPG actor in the combined controller (BASELINE configs[2]).

Golden data (tests/golden/make_golden_combined_real.py, build container): the reference's own ``do_combined_control`` under
``configs/combined_medium_1.json`` with ``get_control`` = its state-vector function -> float32 -> TimeFeature input -> the tensors of
``pretrained_models/ddpg_medium1_extended/policy.pt`` in torch fp32 -> 5 tanh; every policy evaluation's input vector and jerk recorded.
"""
import numpy as np
import pytest

from conftest import load_golden
from test_combined import _apply_settings

ACC_SLOTS = [0, 4, 8, 12]          # the other vehicles' accelerations in the 21-vector (dqn.py:399-400)


def _golden():
    g = load_golden("golden_combined_real.npz")
    pkg = _apply_settings(g)
    return g, pkg


def test_exported_actor_files():
    from rl_mpc_lanemerging_amd import actor
    for name in actor.PRETRAINED:
        w = actor.load_weights(name)
        assert w["w0"].shape == (400, 21) and w["w1"].shape == (300, 400) and w["w2"].shape == (1, 300)
        assert w["w0"].dtype == np.float32 and w["tanh_scale"] == 5.0 and w["tanh_mean"] == 0.0
        assert np.abs(w["w1"]).max() > 0.01                                        # trained, not an initialisation
    assert actor.weights_path("runs/ddpg_medium1_extended") == actor.weights_path("medium1")
    with pytest.raises(FileNotFoundError):
        actor.weights_path("runs/ddpg_medium9_extended")


def test_host_state_vector_and_network_reproduce_the_reference(restore_settings):
    """CPU: the host twin of k_policy_features gives the recorded input vectors bit for bit (first evaluation of every state: all 21 entries;
    later evaluations, rolled out with the oracle's predictor: all but the accelerations, which the oracle's predictor does not return), and
    the exported tensors give the recorded jerks."""
    from rl_mpc_lanemerging_amd import _capi, actor
    from rl_mpc_lanemerging_amd.combined import get_ego_speed_from_jerk
    from oracle import st_oracle as orc
    g, pkg = _golden()
    S = pkg.Settings
    op = orc.OrcParams.from_dict(_capi.Params.from_settings(S).as_dict())
    n = g["ego"].shape[0]
    later = 0
    for i in range(n):
        k = int(g["k_count"][i])
        v = actor.state_vector_host(S, g["ego"][i, :4], g["other_x"][i, :k], g["other_v"][i, :k], g["other_a"][i, :k], int(g["evals0"][i]))
        assert np.array_equal(v, g["vectors"][i, 0]), i
        if i % 4:                     # every fourth state's whole rollout
            continue
        st_ = orc.make_state(*g["ego"][i, :4], g["other_x"][i, :k], g["other_v"][i, :k])
        for j in range(1, int(g["n_evals"][i])):
            sel = get_ego_speed_from_jerk(st_.ego_v, st_.ego_a, float(g["jerks"][i, j - 1]))
            st_, _ = orc.predict_with_ego(op, st_, sel, S.TICK_LENGTH, S.COMBINATION_MIN_DISTANCE)
            xs, vs = orc.state_lists(st_)
            v = actor.state_vector_host(S, (st_.ego_x, st_.ego_y, st_.ego_v, st_.ego_a), xs, vs, [0.0] * k, int(g["evals0"][i]) + j)
            keep = [q for q in range(21) if q not in ACC_SLOTS]
            assert np.array_equal(v[keep], g["vectors"][i, j][keep]), (i, j)
            later += 1
    assert later > 1000
    w = actor.load_weights(str(g["actor"]))
    live = ~np.isnan(g["jerks"])
    out = actor.forward_host(w, g["vectors"][live])
    assert np.abs(out - g["jerks"][live]).max() < 2e-5              # float32 GEMMs in another summation order
    assert live.sum() == g["n_evals"].sum()


def test_reference_decisions_follow_from_the_recorded_jerks(restore_settings):
    """CPU: the decision tree on the oracle's predictor + solver, fed the recorded jerks, gives the reference's decisions (the same check as
    test_combined's, on the states and actions of the real actor)."""
    from rl_mpc_lanemerging_amd import _capi
    from rl_mpc_lanemerging_amd.combined import get_ego_speed_from_jerk
    from oracle import st_oracle as orc
    g, pkg = _golden()
    S = pkg.Settings
    p = _capi.Params.from_settings(S)
    op = orc.OrcParams.from_dict(p.as_dict())
    n = g["ego"].shape[0]
    reason = np.zeros(n, dtype=np.int32)
    probe = []
    for i in range(n):
        k = int(g["k_count"][i])
        st_ = orc.make_state(*g["ego"][i, :4], g["other_x"][i, :k], g["other_v"][i, :k])
        crash, test_state, j = False, None, 0
        while not (crash or j >= max(S.ROLLOUT_LENGTH, 1)):
            j += 1
            sel = get_ego_speed_from_jerk(st_.ego_v, st_.ego_a, float(g["jerks"][i, j - 1]))
            st_, crash = orc.predict_with_ego(op, st_, sel, S.TICK_LENGTH, S.COMBINATION_MIN_DISTANCE)
            if j == S.ST_TEST_ROLLOUTS:
                test_state = st_
            if st_.ego_x > S.STOP_X:
                break
        assert j == g["n_evals"][i]
        if test_state is None:
            test_state = st_
        if crash:
            reason[i] = 1
        else:
            probe.append((i, test_state))
    ego = np.array([[t.ego_x, t.ego_y, t.ego_v, t.ego_a, orc.ego_s(t.ego_x, t.ego_y)] for _, t in probe])
    kc = np.array([t.k for _, t in probe], dtype=np.int32)
    ox = np.zeros((len(probe), 8)); ov = np.zeros((len(probe), 8))
    for r, (_, t) in enumerate(probe):
        xs, vs = orc.state_lists(t)
        ox[r, :t.k] = xs; ov[r, :t.k] = vs
    res = orc.solve_batch(op, ego, kc, ox, ov, solver="layered", nthreads=8)
    for r, (i, _) in enumerate(probe):
        if res["crash"][r]:
            reason[i] = 3
    assert np.array_equal(reason, g["reason"])
    assert (g["reason"] == 1).sum() > 40 and (g["reason"] == 3).sum() > 20 and (g["reason"] == 0).sum() > 1000


def _device_inputs(g, dev):
    import torch
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    return t(g["ego"]), t(g["k_count"]), t(g["other_x"]), t(g["other_v"]), t(g["other_a"])


@pytest.mark.gpu
def test_gpu_features_and_decisions_with_recorded_jerks(gpu_ctx, restore_settings):
    """GPU, bit-exact part: k_policy_features gives the reference's input vector at every policy evaluation of every rollout (all 21 entries,
    incl. the accelerations the device predictor hands on and the evaluation counters), and with the recorded jerks fed back the decisions
    and reasons are the reference's -- dense and sparse controller solve alike."""
    import torch
    from rl_mpc_lanemerging_amd import _capi, actor, combined
    g, pkg = _golden()
    S = pkg.Settings
    dev = torch.device("cuda", torch.cuda.current_device())
    params = _capi.Params.from_settings(S)
    d_ego, d_k, d_ox, d_ov, d_oa = _device_inputs(g, dev)
    n = d_ego.shape[0]
    jerks = torch.as_tensor(np.nan_to_num(g["jerks"]), device=dev)
    n_evals = g["n_evals"]
    results = {}
    for sparse in (True, False):
        pol = actor.DDPGActor(str(g["actor"]), n, gpu_ctx, S, dev)
        pol.evals.copy_(torch.as_tensor(g["evals0"], device=dev))
        seen = []

        def policy(step, cur_ego4, k_, cur_ox, cur_ov, cur_oa):
            seen.append(pol.features(step, cur_ego4, k_, cur_ox, cur_ov, cur_oa).cpu().numpy().copy())
            return jerks[:, step - 1].contiguous()
        cfg = _capi.CombinedCfg.from_settings(S, sparse_control=sparse)
        d = combined.decide_batch_device(gpu_ctx, params, cfg, d_ego, d_k, d_ox, d_ov, policy, None, torch.cuda.current_stream().cuda_stream, d_oa=d_oa)
        torch.cuda.synchronize()
        gpu_ctx.check_error()
        for step, v in enumerate(seen):
            rows = n_evals > step
            assert np.array_equal(v[rows], g["vectors"][rows, step]), ("features", step)
        assert np.array_equal(pol.evals.cpu().numpy(), g["evals0"] + n_evals)
        results[sparse] = {q: d[q].cpu().numpy() for q in ("takeover", "reason", "speed")}
        assert np.array_equal(results[sparse]["reason"], g["reason"])
        assert np.array_equal(results[sparse]["takeover"], g["takeover"])
    assert np.array_equal(results[True]["speed"], results[False]["speed"])          # the same commands, whichever states the controller was solved for
    dec, solves = gpu_ctx.combined_counts(reset=True)
    assert solves >= n + int(g["takeover"].sum())


@pytest.mark.gpu
def test_gpu_actor_network_and_its_unpinned_inputs(gpu_ctx, restore_settings):
    """GPU, floating-point part: the network evaluated on the device -- the fused MFMA kernel (k_actor_eval) and the PyTorch-ROCm engine --
    against the recorded torch-CPU jerks of the reference-side run, and how many of the reference's decisions move (a) with the fused kernel,
    (b) with torch float32 on the device, (c) in float64, (d) without the TimeFeature input -- the one input that restates an absent library."""
    import torch
    from rl_mpc_lanemerging_amd import _capi, actor, combined
    g, pkg = _golden()
    S = pkg.Settings
    dev = torch.device("cuda", torch.cuda.current_device())
    params = _capi.Params.from_settings(S)
    cfg = _capi.CombinedCfg.from_settings(S)
    d_ego, d_k, d_ox, d_ov, d_oa = _device_inputs(g, dev)
    n = d_ego.shape[0]
    live = ~np.isnan(g["jerks"])
    moved, first = {}, {}
    for label, kw in (("hip", dict(engine="hip")), ("torch_fp32", dict(engine="torch")), ("torch_fp64", dict(engine="torch", dtype=torch.float64)),
                      ("hip_no_time_feature", dict(engine="hip", time_feature=False))):
        pol = actor.DDPGActor(str(g["actor"]), n, gpu_ctx, S, dev, **kw)
        pol.evals.copy_(torch.as_tensor(g["evals0"], device=dev))
        if pol.engine == "torch" and label != "hip_no_time_feature":       # the network alone, on every recorded input vector
            out = pol.forward(torch.as_tensor(g["vectors"][live], device=dev)).cpu().numpy()
            assert np.abs(out - g["jerks"][live]).max() < 5e-5, label
        pol.keep_features = True
        d = combined.decide_batch_device(gpu_ctx, params, cfg, d_ego, d_k, d_ox, d_ov, pol, None, torch.cuda.current_stream().cuda_stream, d_oa=d_oa)
        torch.cuda.synchronize()
        gpu_ctx.check_error()
        assert pol.engine == ("hip" if label.startswith("hip") else "torch")
        reason = d["reason"].cpu().numpy()
        moved[label] = int((reason != g["reason"]).sum())
        first[label] = d["first_action"].cpu().numpy()
        if label in ("hip", "torch_fp32"):
            assert np.abs(first[label] - g["jerks"][:, 0]).max() < 5e-5, label     # same inputs (bit-exact), float32 sums in another order
        assert np.array_equal(pol.evals.cpu().numpy() >= g["evals0"] + 1, np.ones(n, bool))
    assert np.abs(first["hip"] - first["torch_fp32"]).max() < 5e-5
    print("decisions that differ from the reference's (of %d): %s" % (n, moved))
    for label in ("hip", "torch_fp32", "torch_fp64"):
        assert moved[label] <= n // 200, (label, moved)                    # float32 summation order can move a borderline rollout; nothing more
    assert moved["hip_no_time_feature"] <= n // 10                         # how much hangs on the restated TimeFeature input (reported)


@pytest.mark.gpu
def test_gpu_fused_actor_kernel_against_torch_fp32(gpu_ctx, restore_settings):
    """k_actor_eval alone: for every exported actor, random states (all vehicle counts, N not a multiple of the 32-state tile): its input
    vectors equal k_policy_features' bit for bit, its jerks equal the PyTorch float32 evaluation of the same network on those inputs within
    float32 rounding of another summation order, and a re-run gives the same bits; argument checks of the entry."""
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, actor, synth
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    S = pkg.Settings
    n, K = 1000 + 7, 8
    ego, kc, ox, ov = synth.generate_states(n, k=7, kmax=K, seed=12, vary_k=True)
    rng = np.random.default_rng(13)
    oa = rng.uniform(-4.0, 2.0, ox.shape)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    d_ego4, d_k, d_ox, d_ov, d_oa = t(ego[:, :4]), t(kc), t(ox), t(ov), t(oa)
    evals0 = rng.integers(0, 900, n).astype(np.int32)
    for name in actor.PRETRAINED:
        hip = actor.DDPGActor(name, n, gpu_ctx, S, dev, engine="hip")
        ref = actor.DDPGActor(name, n, gpu_ctx, S, dev, engine="torch")
        hip.keep_features = True
        outs = []
        for _ in range(2):
            hip.evals.copy_(t(evals0))
            outs.append(hip(1, d_ego4, d_k, d_ox, d_ov, d_oa).clone())
        ref.evals.copy_(t(evals0))
        want = ref(1, d_ego4, d_k, d_ox, d_ov, d_oa)
        torch.cuda.synchronize()
        assert torch.equal(hip.feat, ref.feat), name
        assert torch.equal(outs[0], outs[1]), name
        assert torch.equal(hip.evals, ref.evals)
        err = (outs[0] - want).abs().max().item()
        assert err < 5e-5, (name, err)
        assert outs[0].abs().max().item() <= 5.0 and outs[0].std().item() > 0.5        # a trained policy: uses its range
    fc = _capi.FeaturesCfg.from_settings(S, time_feature=False)                          # 20 inputs: not this actor's width
    with pytest.raises(RuntimeError):
        gpu_ctx.actor_eval_device(hip.handle, fc, n, K, 1, d_ego4.data_ptr(), d_k.data_ptr(), d_ox.data_ptr(), d_ov.data_ptr(), 0, 0, 0, 21, hip.jerk.data_ptr(), 0)
    with pytest.raises(RuntimeError):
        gpu_ctx.actor_create(dict(w0=np.zeros((8, 40), np.float32), b0=np.zeros(8, np.float32), w1=np.zeros((8, 8), np.float32), b1=np.zeros(8, np.float32),
                                  w2=np.zeros((1, 8), np.float32), b2=np.zeros(1, np.float32), tanh_scale=1.0, tanh_mean=0.0))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [dict(), dict(CARS_AHEAD=1, CARS_BEHIND=3), dict(USE_ACCELERATION_OF_OTHER_CARS=False), dict(USE_SPEED_DIFFERENCE=False),
                                   dict(NORMALIZE_VECTOR_INPUT=False), dict(CARS_AHEAD=3, CARS_BEHIND=0, USE_ACCELERATION_OF_OTHER_CARS=False, NORMALIZE_VECTOR_INPUT=False)])
def test_gpu_policy_features_follow_every_flag(flags, gpu_ctx, restore_settings):
    """k_policy_features against its host twin (the reference's dqn.get_state_vector_from_base_state, statement by statement) for the flag
    combinations the reference's function has (dqn.py:390-446), incl. vehicle lists that are not sorted and states without vehicles; with and
    without the TimeFeature input; and the entry's argument checks."""
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, actor, synth
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(flags)
    S = pkg.Settings
    n, K = 300, 8
    ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=K, seed=9, vary_k=True)
    rng = np.random.default_rng(10)
    oa = rng.uniform(-4.0, 2.0, ox.shape)
    for i in range(0, n, 5):                                   # unsorted lists: the reference takes vehicles in list order, whatever it is
        kk = int(kc[i])
        perm = rng.permutation(kk)
        ox[i, :kk], ov[i, :kk], oa[i, :kk] = ox[i, perm], ov[i, perm], oa[i, perm]
    evals0 = rng.integers(0, 900, n).astype(np.int32)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    d_ego4, d_k, d_ox, d_ov, d_oa = t(ego[:, :4]), t(kc), t(ox), t(ov), t(oa)
    for tf in (True, False):
        fc = _capi.FeaturesCfg.from_settings(S, time_feature=tf)
        flen = (fc.cars_ahead + fc.cars_behind) * (4 if fc.use_acceleration else 3) + 4 + (1 if tf else 0)
        d_evals = t(evals0.copy())
        feat = torch.full((n, flen + 3), -7.0, dtype=torch.float32, device=dev)
        gpu_ctx.policy_features_device(fc, n, K, 1, d_ego4.data_ptr(), d_k.data_ptr(), d_ox.data_ptr(), d_ov.data_ptr(), d_oa.data_ptr(),
                                       d_evals.data_ptr() if tf else 0, feat.data_ptr(), flen + 3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = feat.cpu().numpy()
        assert (got[:, flen:] == -7.0).all()                   # nothing written past the vector
        for i in range(n):
            kk = int(kc[i])
            want = actor.state_vector_host(S, ego[i, :4], ox[i, :kk], ov[i, :kk], oa[i, :kk], int(evals0[i]) if tf else None)
            assert np.array_equal(got[i, :flen], want), (i, tf)
        if tf:
            assert np.array_equal(d_evals.cpu().numpy(), evals0 + 1)
    fc = _capi.FeaturesCfg.from_settings(S)
    with pytest.raises(RuntimeError):                          # stride shorter than the vector
        gpu_ctx.policy_features_device(fc, n, K, 1, d_ego4.data_ptr(), d_k.data_ptr(), d_ox.data_ptr(), d_ov.data_ptr(), 0, d_evals.data_ptr(), feat.data_ptr(), 3, 0)
    with pytest.raises(RuntimeError):                          # the time feature needs its counters
        gpu_ctx.policy_features_device(fc, n, K, 1, d_ego4.data_ptr(), d_k.data_ptr(), d_ox.data_ptr(), d_ov.data_ptr(), 0, 0, feat.data_ptr(), 64, 0)


@pytest.mark.gpu
def test_gpu_sparse_control_with_no_takeover_at_all(gpu_ctx, restore_settings):
    """A batch in which no decision calls the controller (a policy that brakes gently far from any vehicle): the sparse path solves nothing,
    keeps every policy command, and the counters say so."""
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, combined, combined_bench
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(combined_bench.COMBINED_MEDIUM_1)
    S = pkg.Settings
    n, K = 130, 8
    dev = torch.device("cuda", torch.cuda.current_device())
    ego = np.zeros((n, 5)); ego[:, 0] = np.linspace(-220.0, -150.0, n); ego[:, 1] = 20.0; ego[:, 2] = 10.0
    from rl_mpc_lanemerging_amd import control
    ego[:, 4] = [control.get_ego_s((x, y)) for x, y in ego[:, :2]]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    d_ego, d_k, d_ox, d_ov = t(ego), t(np.zeros(n, np.int32)), t(np.zeros((n, K))), t(np.zeros((n, K)))
    policy = lambda step, e4, k_, x_, v_, a_: torch.full((n,), -0.5, dtype=torch.float64, device=dev)
    gpu_ctx.combined_counts(reset=True)
    d = combined.decide_batch_device(gpu_ctx, _capi.Params.from_settings(S), _capi.CombinedCfg.from_settings(S, sparse_control=True), d_ego, d_k, d_ox, d_ov, policy, None,
                                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    gpu_ctx.check_error()
    assert int(d["takeover"].sum()) == 0 and int(d["reason"].abs().sum()) == 0
    want = [combined.get_ego_speed_from_jerk(10.0, 0.0, -0.5)] * n
    assert np.array_equal(d["speed"].cpu().numpy(), np.array(want))
    assert gpu_ctx.combined_counts() == (n, 0)


@pytest.mark.gpu
def test_gpu_sparse_and_dense_controller_solve_agree_on_a_bench_batch(gpu_ctx, restore_settings):
    """configs[2]'s bench batch (4096 states, pretrained ddpg_medium1, fused kernel): the sparse controller solve -- only the states whose decision
    calls st.do_st_control -- and the dense one give the same takeover flags, reasons and commanded speeds, bit for bit; the counters show the saving."""
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, actor, combined, combined_bench
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(combined_bench.COMBINED_MEDIUM_1)
    S = pkg.Settings
    n = 4096
    ego, kc, ox, ov, evals0 = combined_bench.bench_states(n, 3000, S)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    d_ego, d_k, d_ox, d_ov, d_e0 = t(ego), t(kc), t(ox), t(ov), t(evals0)
    params = _capi.Params.from_settings(S)
    pol = actor.DDPGActor(combined_bench.COMBINED_MEDIUM_1_ACTOR, n, gpu_ctx, S, dev)
    res = {}
    for sparse in (True, False):
        pol.evals.copy_(d_e0)
        gpu_ctx.combined_counts(reset=True)
        d = combined.decide_batch_device(gpu_ctx, params, _capi.CombinedCfg.from_settings(S, sparse_control=sparse), d_ego, d_k, d_ox, d_ov, pol, None,
                                         torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        gpu_ctx.check_error()
        res[sparse] = ({q: d[q].cpu().numpy() for q in ("takeover", "reason", "speed", "first_action")}, gpu_ctx.combined_counts())
    for q in ("takeover", "reason", "first_action"):
        assert np.array_equal(res[True][0][q], res[False][0][q]), q
    assert np.array_equal(res[True][0]["speed"].view(np.uint64), res[False][0]["speed"].view(np.uint64))
    takeovers = int(res[True][0]["takeover"].sum())
    assert 100 < takeovers < n // 4
    assert res[True][1] == (n, takeovers) and res[False][1] == (n, n)
