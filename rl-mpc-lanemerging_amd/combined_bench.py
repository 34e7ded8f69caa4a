"""``bench.py --workload combined``: one simulator tick of the RL + MPC combined controller (reference
``configs/combined_medium_1.json``: EVALUATE_COMBINED_DDPG, dqn.py:117-200) for N states, everything on the device.

The reference's actor is a pretrained DDPG network loaded through the ``all`` library (not available); this
workload uses a STAND-IN of the same shape -- 21 -> 400 -> 300 -> 1, ReLU, tanh x MAXIMUM_POSITIVE_JERK (ddpg.py:29-34,83-87)
-- with seeded random weights on torch-ROCm, fed the reference's state vector (dqn.get_state_vector_from_base_state,
dqn.py:400-445: two cars ahead, two behind, (acceleration, speed difference, gap, 1) each, then ego speed, acceleration and
position, normalised, one zero pad).  A tick = ROLLOUT_LENGTH policy evaluations + rollout steps, the feasibility probe solve
of the rolled-out state, the controller solve (lattice search + QP re-sampling) of the start state and the decision rules.
"""
import json
import os
import time

import numpy as np

COMBINED_MEDIUM_1 = dict(ROLLOUT_LENGTH=5, ST_TEST_ROLLOUTS=5, LIMIT_DQN_SPEED=False, TEST_ST_STRICTLY_BETTER=False, TEST_ROLLOUT_STATE=True,
                         CHECK_ROLLOUT_CRASH=True, COMBINATION_MIN_DISTANCE=5.1, STOP_X=65, REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED=False)


def state_vector(torch, S, ego4, k, ox, ov, oa):
    """dqn.get_state_vector_from_base_state (CARS_AHEAD = CARS_BEHIND = 2, USE_ACCELERATION_OF_OTHER_CARS, USE_SPEED_DIFFERENCE,
    NORMALIZE_VECTOR_INPUT with SENSOR_RADIUS 125), batched: [N, 21]."""
    n, K = ox.shape
    idx = torch.arange(K, device=ox.device)[None, :]
    valid = idx < k[:, None]
    dx = ox - ego4[:, 0:1]
    feat = torch.stack([oa / 9.0, (ov - ego4[:, 2:3]) / S.MAX_SPEED, dx / 125.0, torch.ones_like(ox)], dim=2)      # [N, K, 4]
    big = torch.full_like(dx, 1e9)
    ahead = valid & (dx > 0)
    behind = valid & ~(dx > 0)
    out = []
    for mask, key in ((ahead, dx), (behind, -dx)):        # nearest first (front list reversed, back list in order)
        order = torch.where(mask, key, big).argsort(dim=1)[:, :2]
        picked = torch.gather(feat, 1, order[:, :, None].expand(-1, -1, 4))
        ok = torch.gather(mask, 1, order)
        out.append(torch.where(ok[:, :, None], picked, torch.zeros_like(picked)).reshape(n, 8))
    ego = torch.stack([ego4[:, 2] / S.MAX_SPEED, ego4[:, 3] / 9.0, ego4[:, 0] / 300.0, ego4[:, 1] / 100.0, torch.zeros_like(ego4[:, 0])], dim=1)
    return torch.cat(out + [ego], dim=1)


def make_stand_in_policy(torch, S, dev, seed=1234):
    """STAND-IN for the reference's DDPG actor (ddpg.py:29-34,83-87): 21 -> 400 -> 300 -> 1, ReLU, tanh x MAXIMUM_POSITIVE_JERK, seeded random
    weights; returns the ``policy(step, ego4, k, ox, ov, oa) -> jerk[N]`` callback ``combined.decide_batch_device`` takes."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    actor = torch.nn.Sequential(torch.nn.Linear(21, 400), torch.nn.ReLU(), torch.nn.Linear(400, 300), torch.nn.ReLU(), torch.nn.Linear(300, 1))
    with torch.no_grad():
        for prm in actor.parameters():
            prm.copy_(torch.empty_like(prm).uniform_(-0.3, 0.3, generator=g))
    actor = actor.to(dev).double()
    jmax = float(S.MAXIMUM_POSITIVE_JERK)

    def policy(step, cur_ego4, k_, cur_ox, cur_ov, cur_oa):
        with torch.no_grad():
            return jmax * torch.tanh(actor(state_vector(torch, S, cur_ego4, k_, cur_ox, cur_ov, cur_oa)).squeeze(1))
    return policy


def run(args, rank, world, dev, dist):
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, combined, synth
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(COMBINED_MEDIUM_1)
    S = pkg.Settings
    params = _capi.Params.from_settings(S)
    cfg = _capi.CombinedCfg.from_settings(S)
    n = args.episodes if args.episodes > 0 else 4096
    K, Kmax = 6, 8
    ego, kc, ox, ov = synth.generate_states(n, k=K, kmax=Kmax, seed=3000 + rank, dt=S.T_DISCRETIZATION)
    ctx = _capi.Context(dev.index or 0)
    d_ego, d_k = torch.as_tensor(ego, device=dev), torch.as_tensor(kc, device=dev)
    d_ox, d_ov = torch.as_tensor(ox, device=dev), torch.as_tensor(ov, device=dev)
    policy = make_stand_in_policy(torch, S, dev)

    def tick():
        return combined.decide_batch_device(ctx, params, cfg, d_ego, d_k, d_ox, d_ov, policy, None, torch.cuda.current_stream().cuda_stream)

    for _ in range(args.warmup):
        d = tick()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d = tick()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    reason = d["reason"].cpu().numpy()
    out = {"metric": "combined RL+MPC controller ticks/sec (configs/combined_medium_1.json decision logic, stand-in actor)",
           "value": n * world * args.steps / elapsed, "unit": "ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": "N=%d states/GPU, one tick of dqn.RLAgent.do_combined_control: %d-step policy rollout, feasibility probe solve, "
                                  "controller solve (H=%d, S=%d) + QP re-sampling, decision; actor = STAND-IN 21-400-300-1 MLP with seeded random "
                                  "weights (the reference's pretrained DDPG actor cannot be loaded here)" % (n, max(int(S.ROLLOUT_LENGTH), 1), _capi.num_t(params), _capi.num_s(params, 0.0)),
                      "episodes_per_gpu": n},
           "decisions": {"policy_kept": int((reason == 0).sum()), "crash_predicted": int((reason == 1).sum()), "too_fast": int((reason == 2).sum()),
                         "probe_rejected": int((reason == 3).sum()), "st_better": int((reason == 4).sum())}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["parity_note"] = "decision parity with the reference's logic is pinned by tests/test_combined.py (golden_combined*.npz); this workload has no CPU baseline (stand-in actor)"
    return out
