"""``bench.py --workload combined``: one simulator tick of the RL + MPC combined controller (BASELINE configs[2] = the reference's
``configs/combined_medium_1.json``: EVALUATE_COMBINED_DDPG, dqn.py:117-200) for N states, everything on the device.

The actor is the reference's own pretrained network of that config (``MODEL_NAME runs/ddpg_medium1_extended`` =
``pretrained_models/ddpg_medium1_extended/policy.pt``, its tensors exported as data by data/make_actor_weights.py): 21 -> 400 ->
300 -> 1, ReLU, 5 x tanh (ddpg.py:29-41,83-87), evaluated in float32 on PyTorch-ROCm and fed by ``k_policy_features`` (``actor.DDPGActor``).
A tick = ROLLOUT_LENGTH policy evaluations + rollout steps, the feasibility probe solve of the rolled-out state, the controller solve
(lattice search + QP re-sampling) of the start states whose decision hands control over, and the decision rules.
"""
import json
import os
import time

import numpy as np

COMBINED_MEDIUM_1 = dict(ROLLOUT_LENGTH=5, ST_TEST_ROLLOUTS=5, LIMIT_DQN_SPEED=False, TEST_ST_STRICTLY_BETTER=False, TEST_ROLLOUT_STATE=True,
                         CHECK_ROLLOUT_CRASH=True, COMBINATION_MIN_DISTANCE=5.1, STOP_X=65, REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED=False)
COMBINED_MEDIUM_1_ACTOR = "runs/ddpg_medium1_extended"        # configs/combined_medium_1.json:4
COMBINED_MEDIUM_1_TRAFFIC = dict(BASE_TRAFFIC_INTERVAL=1.8, OTHER_CAR_SPEED=7.0)      # configs/combined_medium_1.json:7-8


def bench_states(n, seed, S):
    """Seeded synthetic tick states in the range the trained policy drives in (saved_data.csv: mean speed 10-12 m/s, |a| mostly < 1),
    half of them on the approach and in the merge zone where the decisions are made; vehicles cruising (acceleration 0)."""
    from rl_mpc_lanemerging_amd import control, synth
    ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=seed, dt=S.T_DISCRETIZATION)
    rng = np.random.default_rng(seed + 1)
    ego[:, 0] = rng.uniform(-200.0, 55.0, n)
    ego[n // 2:, 0] = rng.uniform(-100.0, 20.0, n - n // 2)
    ego[:, 1] = synth.road_y(ego[:, 0])
    ego[:, 2] = np.clip(rng.normal(12.0, 4.0, n), 0.5, 24.0)
    ego[:, 3] = np.clip(rng.normal(0.0, 1.0, n), -3.0, 3.0)
    ego[:, 4] = [control.get_ego_s((x, y)) for x, y in ego[:, :2]]
    evals0 = rng.integers(0, 5 * 150, n).astype(np.int32)       # policy evaluations since the episode began (TimeFeature): up to 30 s of ticks
    return ego, kc, ox, ov, evals0


def run(args, rank, world, dev, dist):
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, combined, synth
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(COMBINED_MEDIUM_1)
    S = pkg.Settings
    params = _capi.Params.from_settings(S)
    cfg = _capi.CombinedCfg.from_settings(S, sparse_control=True)      # as the reference: the controller solve only where the decision calls it
    n = args.episodes if args.episodes > 0 else 4096
    K, Kmax = 6, 8
    ego, kc, ox, ov, evals0 = bench_states(n, 3000 + rank, S)
    ctx = _capi.Context(dev.index or 0)
    d_ego, d_k = torch.as_tensor(ego, device=dev), torch.as_tensor(kc, device=dev)
    d_ox, d_ov = torch.as_tensor(ox, device=dev), torch.as_tensor(ov, device=dev)
    d_evals0 = torch.as_tensor(evals0, device=dev)
    from rl_mpc_lanemerging_amd import actor as actor_mod

    def timed(engine):
        policy = actor_mod.DDPGActor(COMBINED_MEDIUM_1_ACTOR, n, ctx, S, dev, engine=engine)

        def tick():
            policy.evals.copy_(d_evals0)          # every timed tick is the same tick of the same episodes
            return combined.decide_batch_device(ctx, params, cfg, d_ego, d_k, d_ox, d_ov, policy, None, torch.cuda.current_stream().cuda_stream)

        for _ in range(args.warmup):
            d_ = tick()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            d_ = tick()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, d_

    # BASELINE configs[2] words the actor as "on PyTorch-ROCm": timed that way too, next to the fused kernel the package uses by default
    elapsed_torch, d_torch = timed("torch")
    ctx.combined_counts(reset=True)
    elapsed, d = timed("hip")
    same_decisions = int((d["reason"] != d_torch["reason"]).sum().item())
    reason = d["reason"].cpu().numpy()
    decisions, control_solves = ctx.combined_counts()
    out = {"metric": "combined RL+MPC controller ticks/sec (configs/combined_medium_1.json, pretrained ddpg_medium1 actor)",
           "value": n * world * args.steps / elapsed, "unit": "ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": "N=%d states/GPU, one tick of dqn.RLAgent.do_combined_control: %d-step policy rollout (pretrained ddpg_medium1 actor, "
                                  "21-400-300-1, fp32: state vector + network in one fused MFMA launch, k_actor_eval), feasibility probe solve, controller solve (H=%d, S=%d) + QP "
                                  "re-sampling for the states whose decision hands control over, decision"
                                  % (n, max(int(S.ROLLOUT_LENGTH), 1), _capi.num_t(params), _capi.num_s(params, 0.0)),
                      "episodes_per_gpu": n, "actor": COMBINED_MEDIUM_1_ACTOR, "controller_solves_per_decision": control_solves / max(decisions, 1)},
           "actor_on_pytorch_rocm": {"value": n * world * args.steps / elapsed_torch, "unit": "ticks/s", "ms_per_step": elapsed_torch / args.steps * 1e3,
                                     "note": "same tick with the network as three rocBLAS GEMMs + elementwise kernels on PyTorch-ROCm (inputs from k_policy_features)",
                                     "decisions_that_differ_from_the_fused_kernel": same_decisions},
           "library": {"backend": _capi.backend_info()},
           "decisions": {"policy_kept": int((reason == 0).sum()), "crash_predicted": int((reason == 1).sum()), "too_fast": int((reason == 2).sum()),
                         "probe_rejected": int((reason == 3).sum()), "st_better": int((reason == 4).sum())}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["parity_note"] = ("decisions with this actor are pinned by the suite (test_actor.py) against golden_combined_real.npz (the reference's do_combined_control with the same "
                              "network); the decision logic by test_combined.py; this workload has no CPU baseline leg")
    return out
