#!/usr/bin/env python3
"""bench.py -- MPC (ST lattice) solves/s on MI355X.

Workload (BASELINE.json configs[1]): batched synthetic merge states, 4096 episodes per GPU,
H = 40 time layers, fan-out A = 20-21 (SURVEY 8d mapping: S = 7201 cells), K = 6 neighbours, fp64.
One "step" = one pass of the hot path (traffic prediction -> lattice DP -> path/cost/crash
outputs) over the batch, inputs already resident in HBM.  With --gpus N every rank solves its own
block of episodes and the ranks all-gather the chosen (action, cost) over RCCL.  The per-rank batch is
the SAME for every N (weak scaling, 4096 unless --episodes says otherwise), so the points of a scaling
curve differ in nothing but the rank count; BASELINE configs[3] (65536 episodes over 8 GPUs = 8192 per
rank) is timed as well in every multi-rank run and reported under "config4_shard" (its one-GPU
counterpart is `--episodes 8192`).

Launch: `python bench.py --gpus N` starts the N ranks itself (one process per GPU); under
torchrun (WORLD_SIZE set) it joins as one rank.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6   # fp64 vector peak
# counters bench.py cannot regenerate itself (rocprofv3 --pmc passes, scripts/profile_gpu.sh + scripts/profile_summarize.py): the newest
# round's file that exists.  Each workload entry carries the source hash of the library it was taken from ("csrc_hash"); the bench line says
# "stale": true wherever it quotes such a counter and the running library was built from other sources.
MEASURED_CANDIDATES = [os.path.join(REPO, "profiles", r, "measured.json") for r in ("r6", "r5", "r4", "r3")]
MEASURED = next((m for m in MEASURED_CANDIDATES if os.path.exists(m)), MEASURED_CANDIDATES[0])


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--episodes", type=int, default=0, help="episodes per GPU (default 4096 for every --gpus N)")
    ap.add_argument("--workload", choices=["h40a21", "default", "control", "combined", "episodes"], default="h40a21",
                    help="h40a21: BASELINE workload; default: the reference's own lattice; control: st.do_st_control on the "
                         "reference's lattice (lattice search + QP re-sampling + commanded speed); combined: one tick of the "
                         "RL+MPC combined controller (configs/combined_medium_1.json) with the reference's pretrained ddpg_medium1 actor; episodes: batched merge "
                         "environments with configs/train_moderate_1.json's traffic under that controller (BASELINE configs[4] as a labelled "
                         "throughput demo: the reference has no counterpart)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` object of the headline line (the reference's own lattice, st.do_st_control and "
                    "the combined controller's tick, each a short timed run in the same process after the headline loop)")
    ap.add_argument("--pipelined", type=int, default=2, help="also report the throughput with this many batches in flight (one context, stream and "
                    "output buffers each; 0/1 = skip); the headline value is always one batch at a time")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall-clock target per CPU solver (heap, layered) of the cpu_baseline sample")
    ap.add_argument("--seeds", type=str, default="1000,1,2,3,4", help="state-generator seeds of the seed-median figure (h40a21 / default workloads, one GPU); '' = skip")
    return ap.parse_args()


def cpu_info():
    """(model name, physical cores, logical cpus) of the host."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_phys = min(len(phys), logical) if phys else logical
    # CPU-time quota of the container (cgroup v2 cpu.max / v1 cfs quota): more runnable threads than that are throttled
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = max(1, min(n_phys, int(quota + 0.5))) if quota else max(n_phys, 1)
    return model, max(n_phys, 1), logical, usable, quota


def cpu_baseline(orc, op, ego, kc, ox, ov, seconds):
    """Oracle on the host: binary-heap Dijkstra (the reference's algorithm) and the layered DP, all physical cores,
    one persistent worker per core with its scratch allocated once; the faster is the reported baseline."""
    model, phys_total, logical, phys, quota = cpu_info()      # phys: cores this process can keep busy (host cores capped by the container's CPU quota)
    n = ego.shape[0]
    out = {"cpu_model": model, "physical_cores": phys_total, "logical_cpus": logical, "threads": phys, "cpu_quota": quota}
    c0 = time.perf_counter()
    one = orc.solve_batch(op, ego[:4], kc[:4], ox[:4], ov[:4], solver="heap", nthreads=1)
    t_one = {"heap": (time.perf_counter() - c0) / 4}
    c0 = time.perf_counter()
    orc.solve_batch(op, ego[:4], kc[:4], ox[:4], ov[:4], solver="layered", nthreads=1)
    t_one["layered"] = (time.perf_counter() - c0) / 4
    best, counts = None, {}
    for solver in ("heap", "layered"):
        m = int(min(n, max(4 * phys, seconds * phys / max(t_one[solver], 1e-6))))
        m -= m % max(phys, 1) if m >= 2 * phys else 0
        reps, wall, res = 0, 0.0, None
        while wall < 0.6 * seconds or reps == 0:
            c0 = time.perf_counter()
            res = orc.solve_batch(op, ego[:m], kc[:m], ox[:m], ov[:m], solver=solver, nthreads=phys)
            wall += time.perf_counter() - c0
            reps += 1
        rate = m * reps / wall
        counts[solver] = res
        out[solver] = {"solves_per_s": rate, "per_thread": rate / phys, "single_thread": 1.0 / t_one[solver], "episodes": m,
                       "repeats": reps, "wall_s": wall}
        if best is None or rate > out[best]["solves_per_s"]:
            best = solver
    out["solver"] = best
    return out, counts, one


def secondary_workloads(args, dev, local_rank, np, torch, pkg, _capi, synth):
    """The other BASELINE configs in the driver's one run: configs[0]'s parameters (configs/st_low.json:14-25 = the reference's own lattice, H=18,
    S=3001, fan-out <= 6) as a batched solve with every episode against the oracle, `st.do_st_control` on the same states (lattice search + QP
    re-sampling + commanded speed, st.py:757-783), and configs[2] (configs/combined_medium_1.json: one tick of dqn.RLAgent.do_combined_control,
    dqn.py:117-200, with the reference's pretrained ddpg_medium1 actor).  Each: own context, args.warmup + args.steps steps, timed like the headline
    (synchronise, wall clock).  Returns the `secondary` object; the headline's Settings are restored by the caller."""
    from oracle import st_oracle as orc
    out = {}
    n, K, Kmax = 4096, 6, 8
    steps, warm = max(args.steps, 5), max(args.warmup, 2)
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    S = pkg.Settings
    params = _capi.Params.from_settings(S)
    H = _capi.num_t(params)
    ego, kc, ox, ov = synth.generate_states(n, k=K, kmax=Kmax, seed=1000)
    ctx = _capi.Context(local_rank)
    te, tk, tx, tv = (torch.as_tensor(a_, device=dev) for a_ in (ego, kc, ox, ov))
    o_path = torch.empty((n, H), dtype=torch.int32, device=dev); o_bt = torch.empty(n, dtype=torch.int32, device=dev)
    o_cost = torch.empty(n, dtype=torch.float64, device=dev); o_pd = torch.empty((n, H), dtype=torch.float64, device=dev)
    o_crash = torch.empty(n, dtype=torch.int32, device=dev)
    o_speed = torch.empty(n, dtype=torch.float64, device=dev)
    o_fine = torch.zeros((n, _capi.QP_NMAX), dtype=torch.float64, device=dev); o_fl = torch.zeros(n, dtype=torch.int32, device=dev)

    def timed(fn):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return time.perf_counter() - c0

    def solve():
        ctx.solve_batch_device(params, n, Kmax, te.data_ptr(), tk.data_ptr(), tx.data_ptr(), tv.data_ptr(), o_path.data_ptr(), o_bt.data_ptr(),
                               o_cost.data_ptr(), o_pd.data_ptr(), o_crash.data_ptr(), torch.cuda.current_stream().cuda_stream, 0)
    el = timed(solve)
    c0 = time.perf_counter()
    ref = orc.solve_batch(orc.OrcParams.from_dict(params.as_dict()), ego, kc, ox, ov, solver="layered", nthreads=cpu_info()[3])
    t_orc = time.perf_counter() - c0
    got = {"path_idx": o_path.cpu().numpy(), "best_t": o_bt.cpu().numpy(), "cost": o_cost.cpu().numpy(), "crash": o_crash.cpu().numpy()}
    out["reference_lattice"] = {"metric": "MPC solves/sec (reference default H=%d,S=%d,K=%d; configs/st_low.json parameters)" % (H, _capi.num_s(params, 0.0), K),
                                "value": n * steps / el, "unit": "solves/s", "ms_per_step": el / steps * 1e3, "episodes": n, "steps": steps,
                                "parity_vs_oracle": {"episodes": n, **{k_: bool(np.array_equal(got[k_], ref[k_])) for k_ in got}},
                                "oracle_solves_per_s": n / t_orc}

    def control():
        ctx.st_control_batch_device(params, S.TICK_LENGTH, n, Kmax, te.data_ptr(), tk.data_ptr(), tx.data_ptr(), tv.data_ptr(), o_path.data_ptr(),
                                    o_bt.data_ptr(), o_cost.data_ptr(), o_speed.data_ptr(), o_fine.data_ptr(), o_fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    el = timed(control)
    from oracle import ff_oracle as ff
    from rl_mpc_lanemerging_amd import st as st_mod
    fs = ff.settings(S.MAX_SPEED, S.MAX_POSITIVE_ACCELERATION, S.MAX_NEGATIVE_ACCELERATION, S.MAXIMUM_POSITIVE_JERK, S.MINIMUM_NEGATIVE_JERK, S.CAR_LENGTH)
    mq = 256
    want = np.zeros(mq)
    for i in range(mq):
        bt_i = int(ref["best_t"][i])
        s_seq = st_mod.s_values_for(ego[i, 4], params)[ref["path_idx"][i, :bt_i + 1]]
        x = ff.finer_fit(s_seq, S.TICK_LENGTH, S.T_DISCRETIZATION, ego[i, 2], ego[i, 3], fs)[0]
        want[i] = ego[i, 2] if len(x) <= 1 else (x[1] - x[0]) / S.TICK_LENGTH
    out["st_control"] = {"metric": "st.do_st_control commanded speeds/sec (lattice search + QP re-sampling to the %.1f s tick)" % S.TICK_LENGTH,
                         "value": n * steps / el, "unit": "speeds/s", "ms_per_step": el / steps * 1e3, "episodes": n, "steps": steps,
                         "parity": {"episodes": mq, "path_idx": bool(np.array_equal(o_path.cpu().numpy(), ref["path_idx"])),
                                    "speed_vs_own_qp_oracle": bool(np.array_equal(o_speed[:mq].cpu().numpy(), want)),
                                    "note": "QP solve: parity unpinned (cvxopt absent); the oracle is this repo's restatement (oracle/ff_oracle.c)"}}
    del ctx
    from rl_mpc_lanemerging_amd import combined_bench
    import copy
    a2 = copy.copy(args)
    a2.episodes, a2.steps, a2.warmup, a2.no_cpu_baseline = n, steps, warm, True
    cb = combined_bench.run(a2, 0, 1, dev, None)
    out["combined_tick"] = {"metric": cb["metric"], "value": cb["value"], "unit": cb["unit"], "ms_per_step": cb["ms_per_step"], "episodes": n, "steps": steps,
                            "actor": cb["config"]["actor"], "decisions": cb["decisions"],
                            "controller_solves_per_decision": cb["config"]["controller_solves_per_decision"],
                            "actor_on_pytorch_rocm_ticks_per_s": cb["actor_on_pytorch_rocm"]["value"],
                            "decisions_that_differ_between_the_two_actor_engines": cb["actor_on_pytorch_rocm"]["decisions_that_differ_from_the_fused_kernel"]}
    return out


def run(args):
    import numpy as np
    import torch
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi, sharding, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the solver has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("STMPC_BENCH_FORCE_DIST") == "1"     # the flag exercises the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)
        if dist.get_world_size() != world:
            raise SystemExit("RCCL sees %d ranks, expected %d" % (dist.get_world_size(), world))

    if args.workload == "combined":
        from rl_mpc_lanemerging_amd import combined_bench
        out = combined_bench.run(args, rank, world, dev, dist)
    elif args.workload == "episodes":
        from rl_mpc_lanemerging_amd import episodes_bench
        out = episodes_bench.run(args, rank, world, dev, dist)
    else:
        out = run_solver(args, rank, world, local_rank, dev, dist, np, torch, pkg, _capi, sharding, synth)

    if use_dist:
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio: flush it first so that the JSON line is the last line of stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
    # a fast-but-wrong build must not look like a result: any parity flag that is false fails the run (the line above says which)
    par = out.get("parity_vs_oracle") or {}
    bad = [k for k, v in par.items() if v is False]
    sec = out.get("secondary") or {}
    bad += ["secondary.reference_lattice." + k for k, v in (sec.get("reference_lattice", {}).get("parity_vs_oracle") or {}).items() if v is False]
    bad += ["secondary.st_control." + k for k, v in (sec.get("st_control", {}).get("parity") or {}).items() if v is False]
    if out.get("pipelined") and out["pipelined"].get("outputs_identical_across_buffers") is False:
        bad.append("pipelined.outputs_identical_across_buffers")
    if bad:
        sys.stderr.write("bench.py: PARITY FAILURE against the oracle: %s\n" % ", ".join(bad))
        sys.exit(3)


def run_solver(args, rank, world, local_rank, dev, dist, np, torch, pkg, _capi, sharding, synth):
    use_dist = dist is not None
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    if args.workload == "h40a21":
        pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    params = _capi.Params.from_settings(pkg.Settings)
    H, S_nom = _capi.num_t(params), _capi.num_s(params, 0.0)
    n = args.episodes if args.episodes > 0 else 4096
    K, Kmax = 6, 8
    ego, kc, ox, ov = synth.generate_states(n, k=K, kmax=Kmax, seed=1000 + rank)

    ctx = _capi.Context(local_rank)
    d_ego = torch.as_tensor(ego, device=dev)
    d_k = torch.as_tensor(kc, device=dev)
    d_ox = torch.as_tensor(ox, device=dev)
    d_ov = torch.as_tensor(ov, device=dev)
    d_path = torch.empty((n, H), dtype=torch.int32, device=dev)
    d_bt = torch.empty(n, dtype=torch.int32, device=dev)
    d_cost = torch.empty(n, dtype=torch.float64, device=dev)
    d_pd = torch.empty((n, H), dtype=torch.float64, device=dev)
    d_crash = torch.empty(n, dtype=torch.int32, device=dev)
    gathered = torch.empty((n * world, 2), dtype=torch.float64, device=dev) if use_dist else None
    d_ac = torch.empty((n, 2), dtype=torch.float64, device=dev) if use_dist else None      # fused (action, cost) rows, written by the solver's back-track
    control = args.workload == "control"
    d_speed = torch.empty(n, dtype=torch.float64, device=dev) if control else None
    d_fine = torch.zeros((n, _capi.QP_NMAX), dtype=torch.float64, device=dev) if control else None
    d_fine_len = torch.zeros(n, dtype=torch.int32, device=dev) if control else None

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        if control:
            ctx.st_control_batch_device(params, pkg.Settings.TICK_LENGTH, n, Kmax, d_ego.data_ptr(), d_k.data_ptr(),
                                        d_ox.data_ptr(), d_ov.data_ptr(), d_path.data_ptr(), d_bt.data_ptr(), d_cost.data_ptr(),
                                        d_speed.data_ptr(), d_fine.data_ptr(), d_fine_len.data_ptr(), stream)
        else:
            ctx.solve_batch_device(params, n, Kmax, d_ego.data_ptr(), d_k.data_ptr(), d_ox.data_ptr(), d_ov.data_ptr(),
                                   d_path.data_ptr(), d_bt.data_ptr(), d_cost.data_ptr(), d_pd.data_ptr(),
                                   d_crash.data_ptr(), stream, d_ac.data_ptr() if use_dist else 0)
        if use_dist:
            if control:
                d_ac.copy_(sharding.pack_actions(d_path, d_cost))       # (the controller entry has no fused output)
            sharding.gather_actions(d_ac, world, gathered, force=True)   # the step's one collective, on the buffer the solver wrote

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_begin()
    cpu0 = time.thread_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    cpu1 = time.thread_time()      # CPU time this rank's thread spent ISSUING the steps (launches, the collective's enqueue), not waiting for them
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_end()
    rank_ms = [elapsed / args.steps * 1e3]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, t)
        rank_ms = [float(x) / args.steps * 1e3 for x in allt.cpu()]
        elapsed = float(allt.max().item())
        # every rank must hold every rank's (action, cost): check this rank's own rows of the gathered buffer
        own = gathered[rank * n:(rank + 1) * n]
        assert torch.equal(own[:, 0].to(torch.int32), d_path[:, 1]) and torch.equal(own[:, 1], d_cost), "gather mismatch"

    step()                      # one untimed step outside the profiling window to read the tier statistics
    torch.cuda.synchronize()
    tier_stats = ctx.stats()

    def timed_other_batch(n_, seed):
        """Warm-up + args.steps timed steps of another batch (own inputs and outputs, same context); max over ranks; seconds."""
        e_, k_, x_, v_ = synth.generate_states(n_, k=K, kmax=Kmax, seed=seed)
        te, tk, tx, tv = (torch.as_tensor(a_, device=dev) for a_ in (e_, k_, x_, v_))
        o_path = torch.empty((n_, H), dtype=torch.int32, device=dev); o_bt = torch.empty(n_, dtype=torch.int32, device=dev)
        o_cost = torch.empty(n_, dtype=torch.float64, device=dev); o_pd = torch.empty((n_, H), dtype=torch.float64, device=dev)
        o_crash = torch.empty(n_, dtype=torch.int32, device=dev)
        g_ = torch.empty((n_ * world, 2), dtype=torch.float64, device=dev) if use_dist else None
        ac_ = torch.empty((n_, 2), dtype=torch.float64, device=dev) if use_dist else None

        def step_():
            ctx.solve_batch_device(params, n_, Kmax, te.data_ptr(), tk.data_ptr(), tx.data_ptr(), tv.data_ptr(), o_path.data_ptr(), o_bt.data_ptr(),
                                   o_cost.data_ptr(), o_pd.data_ptr(), o_crash.data_ptr(), torch.cuda.current_stream().cuda_stream,
                                   ac_.data_ptr() if use_dist else 0)
            if use_dist:
                sharding.gather_actions(ac_, world, g_, force=True)
        for _ in range(max(args.warmup, 1)):
            step_()
        barrier()
        c0 = time.perf_counter()
        for _ in range(args.steps):
            step_()
        barrier()
        el = time.perf_counter() - c0
        if use_dist:
            t_ = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            el = float(t_.item())
        return el

    # the headline batch is one draw of the state generator; the step time depends on the draw (its longest searches): median over seeds
    seed_median = None
    seeds = [int(x) for x in args.seeds.split(",") if x.strip()] if args.seeds else []
    if seeds and not use_dist and not control:
        per_seed = {}
        for sd in seeds:
            per_seed[sd] = n * args.steps / (elapsed if sd == 1000 + rank else timed_other_batch(n, sd))
        seed_median = {"seeds": seeds, "solves_per_s": {str(k_): v_ for k_, v_ in per_seed.items()},
                       "value_seed_median": float(np.median(list(per_seed.values()))), "min": min(per_seed.values()), "max": max(per_seed.values())}
    # BASELINE configs[3]: 65536 episodes over 8 GPUs = 8192 per rank, timed in every multi-rank run next to the curve's own per-rank batch
    config4 = None
    if use_dist and not control and n != 8192:
        el4 = timed_other_batch(8192, 2000 + rank)
        config4 = {"episodes_per_gpu": 8192, "episodes_total": 8192 * world, "value": 8192 * world * args.steps / el4, "unit": "solves/s",
                   "ms_per_step": el4 / args.steps * 1e3, "note": "BASELINE configs[3] is this with 8 ranks (65536 episodes); not the headline, whose per-rank batch is the same for every N"}

    pipelined = None
    if args.pipelined > 1 and not use_dist and not control:
        # serving pattern: consecutive batches on separate streams / contexts / output buffers, so that one batch's tail (its
        # last long searches, its wide-window launch) overlaps the next batch's start.  Reported next to the headline, never as it.
        S_ = args.pipelined
        ctxs = [ctx] + [_capi.Context(local_rank) for _ in range(S_ - 1)]
        streams = [torch.cuda.Stream() for _ in range(S_)]
        outs = [(d_path, d_bt, d_cost, d_pd, d_crash)] + [tuple(torch.empty_like(t) for t in (d_path, d_bt, d_cost, d_pd, d_crash)) for _ in range(S_ - 1)]

        def pstep(i):
            j = i % S_
            o = outs[j]
            ctxs[j].solve_batch_device(params, n, Kmax, d_ego.data_ptr(), d_k.data_ptr(), d_ox.data_ptr(), d_ov.data_ptr(), o[0].data_ptr(), o[1].data_ptr(),
                                       o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), streams[j].cuda_stream)
        for i in range(2 * S_):
            pstep(i)
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        for i in range(args.steps):
            pstep(i)
        torch.cuda.synchronize()
        pel = time.perf_counter() - p0
        same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][2], o[2]) for o in outs[1:])
        pipelined = {"batches_in_flight": S_, "value": n * args.steps / pel, "unit": "solves/s", "ms_per_step": pel / args.steps * 1e3,
                     "outputs_identical_across_buffers": bool(same)}

    ms_per_step = elapsed / args.steps * 1e3
    value = n * world * args.steps / elapsed

    measured = {}
    try:
        measured = json.load(open(MEASURED)).get(args.workload, {})
    except (OSError, ValueError):
        pass
    lib_hash = _capi.library_source_hash()
    measured_rel = os.path.relpath(MEASURED, REPO)
    measured_stale = bool(measured) and measured.get("csrc_hash") != lib_hash      # counters from another build of the kernels

    # ---- roofline of the dominant kernel (the LDS lattice-DP kernel), from HIP events on the launch stream
    bytes_per_solve = 140 + 16 + 4 * H        # state in (K=6) + action/cost/best_t + path_idx[H]
    dp_ms = prof["dp_kernel_ms"] / max(prof["launches"], 1)
    achieved_gbs = bytes_per_solve * n / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
    roofline = {"bound": "hbm", "limiting_unit": "VALU instruction issue + LDS round-trip latency of a branchy fp64 DP (see issue, fp64_valu); HBM is reported because BASELINE.json asks for it",
                "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": measured.get("hbm_bytes_per_step"),
                "traffic_source": measured.get("hbm_source"), "traffic_stale": measured_stale if measured.get("hbm_bytes_per_step") else None,
                "kernel": "stmpc::k_solve<true,false,...> (LDS lattice DP; one launch per LDS window tier, summed per step)",
                "kernel_ms": dp_ms, "bytes_per_solve": bytes_per_solve, "bytes_per_launch": bytes_per_solve * n,
                # the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of this command (profiles/rN/<round>_summary.txt,
                # written into measured.json by scripts/profile_summarize.py): must agree with kernel_ms; stale = taken from another build of the library
                "rocprof_kernel_avg_ms": measured.get("dominant_kernel_avg_ms"), "rocprof_kernel": measured.get("dominant_kernel"),
                "rocprof_stale": measured_stale if measured.get("dominant_kernel_avg_ms") else None,
                "note": "algorithmic HBM bytes are %d B/solve (SURVEY 8d): the path is fp64-VALU/LDS-latency bound, not HBM bound; "
                        "see fp64_valu and issue for the bounds that apply" % bytes_per_solve}

    metric = {"h40a21": "MPC solves/sec (H=40,A=21,K=6)", "default": "MPC solves/sec (reference default H=18,S=3001,K=6)",
              "control": "st.do_st_control commanded speeds/sec (reference default H=18,S=3001,K=6, QP re-sampling to the 0.2 s tick)"}[args.workload]
    wl = ("batched synthetic merge states N=%d/GPU (%d total), H=%d, S=%d, fan-out<=21, K=%d, fp64" % (n, n * world, H, S_nom, K)
          if args.workload == "h40a21" else
          "batched synthetic merge states N=%d/GPU, reference default lattice H=%d, S=%d, K=%d, fp64" % (n, H, S_nom, K))
    out = {"metric": metric,
           "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": wl, "episodes_per_gpu": n, "episodes_total": n * world, "H": H, "S": S_nom, "K": K,
                      "collective": ("all_gather(action,cost) 16 B/episode over RCCL, %d ranks" % world) if use_dist else "none",
                      "launcher": "torchrun" if os.environ.get("TORCHELASTIC_RUN_ID") else ("self-spawn" if world > 1 else "single")},
           "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
           # what a rank's host side costs: with N ranks on one node the ranks share the container's CPU quota, and a step is ~10 launches + 1 collective
           "host": {"host_us_per_step": (cpu1 - cpu0) / args.steps * 1e6, "cpus_in_affinity_mask": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                    "container_cpu_quota": cpu_info()[4], "note": "rank 0's figures; thread CPU time of the timed step loop / steps"},
           "rccl_ranks": (dist.get_world_size() if use_dist else 0),
           "library": {"backend": _capi.backend_info(), "csrc_hash": lib_hash, "measured_counters": measured_rel,
                       "measured_counters_csrc_hash": measured.get("csrc_hash"), "measured_counters_stale": measured_stale},
           "roofline": roofline, "device_ms_per_step": prof["solve_ms"] / max(prof["launches"], 1),
           "tiers": {"first_lds_window": int(tier_stats["fast_path"]), "larger_lds_window": int(tier_stats["fallback"] - tier_stats["hbm_tier"]),
                     "hbm_scratch": int(tier_stats["hbm_tier"]), "bound_retries": int(tier_stats["retries"]), "guided_bounds": int(tier_stats.get("guided", 0)),
                     "resume_refused": int(tier_stats.get("resume_refused", 0)),
                     "nodes_expanded_per_solve": (tier_stats["nodes_exact"] + tier_stats["nodes_bound"]) / n}}

    if pipelined:
        out["pipelined"] = pipelined
    if seed_median:
        out["value_seed_median"] = seed_median["value_seed_median"]
        out["seed_sweep"] = seed_median
    if config4:
        out["config4_shard"] = config4
    if control:
        out["stages"] = {"lattice_search_ms": prof["solve_ms"] / max(prof["launches"], 1),
                         "qp_resampling_ms": ms_per_step - prof["solve_ms"] / max(prof["launches"], 1),
                         "note": "lattice search from HIP events inside the library; QP = step time minus that"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_secondary and args.workload == "h40a21" and args.episodes in (0, 4096):
        # the other configs, witnessed by the same run (after the timed headline loop; nothing of the headline depends on it)
        c0 = time.perf_counter()
        try:
            out["secondary"] = secondary_workloads(args, dev, local_rank, np, torch, pkg, _capi, synth)
            out["secondary"]["wall_s"] = time.perf_counter() - c0
        finally:
            pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
            pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import st_oracle as orc
        op = orc.OrcParams.from_dict(params.as_dict())
        base, counts, _ = cpu_baseline(orc, op, ego, kc, ox, ov, args.cpu_seconds)
        best = base["solver"]
        m = base[best]["episodes"]
        ref = counts["layered"]
        mp = counts["layered"]["path_idx"].shape[0]
        got = {"path_idx": d_path[:mp].cpu().numpy(), "best_t": d_bt[:mp].cpu().numpy(), "cost": d_cost[:mp].cpu().numpy(),
               "crash": d_crash[:mp].cpu().numpy()}
        if control:
            got.pop("crash")                                   # the controller entry does not compute the crash probe
        parity = {k: bool(np.array_equal(got[k], ref[k])) for k in got}
        cpu_rate = base[best]["solves_per_s"]
        sample = ("first %d episodes of the same batch x %d repeats, oracle/st_oracle.c %s (%s), %d threads (%s: %d physical cores, container CPU quota %s), %.1f s; "
                  "other solver: %.0f solves/s; single-thread rates: heap %.1f, layered %.1f solves/s"
                  % (m, base[best]["repeats"], "binary-heap Dijkstra, the reference's algorithm" if best == "heap" else "layered DP",
                     best, base["threads"], base["cpu_model"], base["physical_cores"], ("%.0f CPUs" % base["cpu_quota"]) if base["cpu_quota"] else "none", base[best]["wall_s"],
                     base["layered" if best == "heap" else "heap"]["solves_per_s"], base["heap"]["single_thread"], base["layered"]["single_thread"]))
        if control:
            # the QP stage on the host: the oracle's restatement, one episode at a time (single thread)
            from oracle import ff_oracle as ff
            from rl_mpc_lanemerging_amd import st as st_mod
            S_ = pkg.Settings
            fs = ff.settings(S_.MAX_SPEED, S_.MAX_POSITIVE_ACCELERATION, S_.MAX_NEGATIVE_ACCELERATION, S_.MAXIMUM_POSITIVE_JERK,
                             S_.MINIMUM_NEGATIVE_JERK, S_.CAR_LENGTH)
            mq = min(mp, 1024)
            want = np.zeros(mq)
            c0 = time.perf_counter()
            for i in range(mq):
                bt_i = int(ref["best_t"][i])
                s_seq = st_mod.s_values_for(ego[i, 4], params)[ref["path_idx"][i, :bt_i + 1]]
                x = ff.finer_fit(s_seq, S_.TICK_LENGTH, S_.T_DISCRETIZATION, ego[i, 2], ego[i, 3], fs)[0]
                want[i] = ego[i, 2] if len(x) <= 1 else (x[1] - x[0]) / S_.TICK_LENGTH
            qp_per_solve = (time.perf_counter() - c0) / mq / base["threads"]     # as if the QPs were spread over the threads too
            parity["speed_vs_own_qp_oracle"] = bool(np.array_equal(d_speed[:mq].cpu().numpy(), want))
            cpu_rate = 1.0 / (1.0 / cpu_rate + qp_per_solve)
            sample += "; plus the QP stage (oracle/ff_oracle.c, parity-unpinned restatement of cvxopt) timed on 1 thread over %d episodes and divided by the core count" % mq
        out["cpu_baseline"] = {"value": cpu_rate, "unit": "solves/s", "cores": base["threads"], "kind": "port",
                               "solver": best, "reference_algorithm_value": base["heap"]["solves_per_s"],
                               "reference_algorithm_note": "oracle's binary-heap Dijkstra = the algorithm st_cy.pyx runs; `value` is the faster of that and the layered DP", "cpu_model": base["cpu_model"], "host_physical_cores": base["physical_cores"],
                               "logical_cpus": base["logical_cpus"], "container_cpu_quota": base["cpu_quota"],
                               "per_thread": base[best]["per_thread"], "single_thread": base[best]["single_thread"],
                               "sample": sample}
        out["parity_vs_oracle"] = {"episodes": mp, **parity}
        # fp64 work: (1) what the reference's algorithm does (heap Dijkstra: settled nodes, relaxed edges, counted by the oracle on
        # the same states), (2) what the full layered DP would do, (3) what the kernel executed (node counters of this run; candidate
        # evaluations per node from the analysis build's counters in the newest profiles/rN/measured.json, flagged stale when taken from other sources)
        hp, ly = counts["heap"], counts["layered"]
        mh = hp["path_idx"].shape[0]
        ref_flops = (27 * hp["edges"] + 26 * hp["nodes"]) / mh + 40 * K * H
        full_flops = (27 * ly["edges"] + 26 * ly["nodes"] + 6 * K * ly["cells"]) / mp + 40 * K * H
        nodes_exec = (tier_stats["nodes_exact"] + tier_stats["nodes_bound"]) / n
        cand_per_node = measured.get("candidates_per_node")
        exec_flops = (27 * cand_per_node + 26) * nodes_exec + 40 * K * H if cand_per_node else None
        per_s = n / (dp_ms * 1e-3) if dp_ms > 0 else 0.0
        out["fp64_valu"] = {"peak_tflops": FP64_VALU_PEAK_TFLOPS,
                            "reference_algorithm_flops_per_solve": ref_flops,
                            "reference_algorithm_frac": ref_flops * per_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                            "executed_flops_per_solve": exec_flops,
                            "executed_frac": (exec_flops * per_s / 1e12 / FP64_VALU_PEAK_TFLOPS) if exec_flops else None,
                            "full_layered_dp_flops_per_solve": full_flops,
                            "full_layered_dp_frac": full_flops * per_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                            "heap_nodes_per_solve": hp["nodes"] / mh, "heap_edges_per_solve": hp["edges"] / mh,
                            "kernel_nodes_per_solve": nodes_exec, "kernel_candidates_per_node": cand_per_node,
                            "kernel_candidates_per_node_stale": measured_stale if cand_per_node else None,
                            "note": "flops = 27/edge + 26/node (+6*K per touched cell for the full DP) + 40*K*H for the predictor (SURVEY 8d); "
                                    "reference_algorithm_* prices the heap Dijkstra's own node/edge counts, which is the useful work"}
        # the unit that is actually half-busy: instruction issue.  Wave-instructions per step from the committed counter pass (same command,
        # same batch), 4 cycles of a SIMD per VALU wave-instruction, 1024 SIMDs at 2.4 GHz for this run's kernel time.
        vi, si = measured.get("per_step_SQ_INSTS_VALU"), measured.get("per_step_SQ_INSTS_SALU")
        if vi and dp_ms > 0 and n == 4096:
            out["issue"] = {"valu_wave_instructions_per_step": vi, "salu_wave_instructions_per_step": si,
                            "valu_issue_slots_per_step": 1024 * dp_ms * 1e-3 * 2.4e9 / 4.0,
                            "valu_busy_frac": vi * 4.0 / (1024 * dp_ms * 1e-3 * 2.4e9),
                            "stale": measured_stale,
                            "source": "%s (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ..., scripts/profile_gpu.sh; taken from library sources %s), kernel time of this run"
                                      % (measured_rel, measured.get("csrc_hash"))}
    return out


def _spawned(local_rank, args, port):
    os.environ["RANK"] = str(local_rank)
    os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["WORLD_SIZE"] = str(args.gpus)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one disjoint set of CPUs per rank (the ranks' host threads -- launch loop, RCCL proxy -- then do not migrate onto each other); skipped when the
    # process may not run on at least two CPUs per rank
    if hasattr(os, "sched_setaffinity"):
        try:
            cpus = sorted(os.sched_getaffinity(0))
            per = len(cpus) // args.gpus
            if per >= 2:
                os.sched_setaffinity(0, cpus[local_rank * per:(local_rank + 1) * per])
        except OSError:
            pass
    run(args)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: launch the N ranks here, one process per GPU
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("--gpus %d requested but only %d visible; refusing to report a smaller run" % (args.gpus, have))
        import torch.multiprocessing as mp
        port = 29500 + (os.getpid() % 2000)
        mp.spawn(_spawned, args=(args, port), nprocs=args.gpus, join=True)
        return
    run(args)


if __name__ == "__main__":
    main()
