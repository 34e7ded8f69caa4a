"""``bench.py --workload episodes``: BASELINE.json configs[4] as a LABELLED THROUGHPUT DEMO.

configs[4] reads "configs/train_moderate_1.json end-to-end DDPG training with HIP-MPC reward shaping, 8-GPU episode batch".  The
reference has no such thing: its training never calls the solver (merge_gym.py has no ``st`` import, the "ST"
reward of dqn.py:449-460 is a closed-form expression), so nothing here has a reference counterpart and nothing is claimed as parity.
What can be measured is the environment side such a training loop would run on this stack: N merge environments with
train_moderate_1.json's traffic (BASE_TRAFFIC_INTERVAL 1.2 s, OTHER_CAR_SPEED 11 m/s), stepped in lock-step on the device by the
combined RL + MPC controller (policy rollout, feasibility probe solve, controller solve + QP re-sampling, decision,
``combined.decide_batch_device``) with the reference's pretrained actor for that traffic (``runs/ddpg_moderate1_extended``, ``actor.DDPGActor``) in place of a learner.  A step = one simulator tick of
every environment; the value is environment steps per second.
"""
import time

import numpy as np

PRE_TICKS = 60          # 12 s of every episode before the timed ticks: the egos are then 80-250 m along the ramp, the first ones merging
TRAIN_MODERATE_1_ENV = dict(BASE_TRAFFIC_INTERVAL=1.2, OTHER_CAR_SPEED=11.0)       # configs/train_moderate_1.json:7-8 (solver parameters: the shipped defaults)


def run(args, rank, world, dev, dist):
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, combined_bench, episodes
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(combined_bench.COMBINED_MEDIUM_1)
    pkg.apply_overrides(TRAIN_MODERATE_1_ENV)
    S = pkg.Settings
    n = args.episodes if args.episodes > 0 else 4096
    ctx = _capi.Context(dev.index or 0)
    from rl_mpc_lanemerging_amd import actor as actor_mod
    policy = actor_mod.DDPGActor("runs/ddpg_moderate1_extended", n, ctx, S, dev)       # configs/combined_moderate_1.json:4, trained on this traffic (train_moderate_1.json)
    r = episodes.EpisodeRunner(n, seed=5000 + rank, controller="combined", policy=policy, ctx=ctx, kmax=16)
    for _ in range(PRE_TICKS):          # untimed: bring the environments to mid-episode, where decisions are made (the first ticks on the ramp are idle ones)
        r.tick()
    for _ in range(args.warmup):
        r.tick()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.tick()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res = r.result()
    running = float((res["status"] == 0).mean())
    return {"metric": "environment steps/sec of batched merge episodes under the combined RL+MPC controller (configs/train_moderate_1.json traffic; throughput demo)",
            "value": n * world * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "reference_counterpart": False, "library": {"backend": _capi.backend_info()},
            "world": {"ego_route": "lane centre line (scenario.py)" if r.cfg.ego_route_n else "straight lines", "junction_rule": int(r.cfg.yield_overlap)},
            "config": {"workload": "N=%d environments/GPU in lock-step, every tick: planner view, %d-step policy rollout (pretrained ddpg_moderate1 actor), feasibility "
                                   "probe solve, controller solve (H=%d, S=%d) + QP re-sampling, decision, world step (Krauss traffic %.1f s / %.0f m/s)"
                                   % (n, max(int(S.ROLLOUT_LENGTH), 1), _capi.num_t(r.params), _capi.num_s(r.params, 0.0), S.BASE_TRAFFIC_INTERVAL, S.OTHER_CAR_SPEED),
                       "episodes_per_gpu": n, "note": "BASELINE configs[4] has no counterpart in the reference (its training never calls the solver, SURVEY R7): demo only"},
            "environments_still_running_at_end": running, "ticks_per_environment": int(r.ticks_done), "untimed_ticks_before": PRE_TICKS,
            "takeover_share": float(np.nanmean(res["percent_st"]))}
