"""Multi-GPU: merge episodes are independent, so a batch is block-partitioned over ranks
(one process per GPU); the only exchange is one all-gather of the chosen action (next cell)
and its cost, 16 B per episode, over RCCL/xGMI (``torch.distributed`` backend "nccl"), or
gloo on CPU for the suite.  No data-path collective exists anywhere else in the path.
"""
import numpy as np


def shard_bounds(n_total, world_size, rank):
    """Contiguous block partition of ``n_total`` episodes: returns ``(lo, hi)`` of this rank."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack_actions(path_idx, cost):
    """Fuse (action, cost) into one fp64 [n, 2] buffer so a single collective moves both.
    The action (lattice index of the first step, < 2**16) is exactly representable."""
    import torch
    buf = torch.empty((path_idx.shape[0], 2), dtype=torch.float64, device=path_idx.device)
    buf[:, 0] = path_idx[:, 1].to(torch.float64)
    buf[:, 1] = cost
    return buf


def gather_actions(local_buf, world_size, out=None, force=False):
    """All-gather the fused per-episode (action, cost) rows of every rank (equal shard sizes).
    ``force`` issues the collective even for a single rank (used to exercise the RCCL path on one GPU)."""
    import torch
    import torch.distributed as dist
    if world_size == 1 and not force:
        return local_buf
    if out is None:
        out = torch.empty((local_buf.shape[0] * world_size, local_buf.shape[1]), dtype=local_buf.dtype,
                          device=local_buf.device)
    dist.all_gather_into_tensor(out, local_buf)
    return out


class ShardedSolver:
    """Solve a global batch with one rank per GPU: each rank solves its block, then all ranks
    hold every episode's (action, cost).  ``solve_fn(ego, k, ox, ov) -> (path_idx[n,H], cost[n])``
    on host arrays is injectable so the partition/gather logic is testable without a GPU."""

    def __init__(self, rank, world_size, solve_fn):
        self.rank, self.world_size, self.solve_fn = rank, world_size, solve_fn

    def solve_global(self, ego, k_count, other_x, other_v, device="cpu"):
        import torch
        n = ego.shape[0]
        if n % self.world_size:
            raise ValueError("global batch must divide evenly over ranks (pad the batch)")
        lo, hi = shard_bounds(n, self.world_size, self.rank)
        path_idx, cost = self.solve_fn(ego[lo:hi], k_count[lo:hi], other_x[lo:hi], other_v[lo:hi])
        buf = pack_actions(torch.as_tensor(np.ascontiguousarray(path_idx), device=device),
                           torch.as_tensor(np.ascontiguousarray(cost), device=device))
        allbuf = gather_actions(buf, self.world_size)
        return allbuf[:, 0].to(torch.int32).cpu().numpy(), allbuf[:, 1].cpu().numpy()
