"""world_size-2 gloo test of the multi-GPU path's partition + gather logic (runs on CPU)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, H, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rl_mpc_lanemerging_amd import sharding

    def fake_solve(ego, k, ox, ov):        # deterministic function of the inputs, stands in for the HIP solver
        path = np.zeros((ego.shape[0], H), dtype=np.int32)
        path[:, 1] = (np.abs(ego[:, 0]) * 7).astype(np.int32) % 5000
        return path, ego[:, 2] * 3.0 + k

    rng = np.random.default_rng(0)
    ego = rng.uniform(-100, 100, (n, 5)); k = rng.integers(0, 7, n).astype(np.int32)
    ox = rng.uniform(-100, 100, (n, 8)); ov = rng.uniform(0, 20, (n, 8))
    solver = sharding.ShardedSolver(rank, world, fake_solve)
    act, cost = solver.solve_global(ego, k, ox, ov)
    exp_path, exp_cost = fake_solve(ego, k, ox, ov)
    ok = np.array_equal(act, exp_path[:, 1]) and np.array_equal(cost, exp_cost)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from rl_mpc_lanemerging_amd import sharding
    for n in (0, 1, 7, 64, 65536, 4097):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 256, 18, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
