"""Multi-GPU plumbing for the grouping path: one process per GPU, `torch.distributed` over
RCCL/xGMI (backend "nccl" on ROCm) -- or gloo in the CPU tests.

The path shards naturally: image batches split by rank for the embedding,
row blocks of the N x N work for distance / re-rank / eps / region query.  The exchanges are
small: one all-gather of the embeddings, all-gathers of the rank lists / sparse V / V_qe /
edge lists, an all-reduce of the eps histogram.  Only list-style collectives that both RCCL
and gloo implement are used, so the same code runs in the gloo tests.
"""
import os

import torch


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun) -> (rank, world, group)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, None
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, dist.group.WORLD


def shard_bounds(n, rank, world):
    """contiguous block of `n` items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(t, group):
    """all-gather equally sized row blocks [r, ...] -> [world*r, ...] (rank order)."""
    if group is None:
        return t
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    t = t.contiguous()
    parts = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(parts, t, group=group)
    return torch.cat(parts, dim=0)


def gather_varlen(t, group):
    """all-gather row blocks of different lengths [n_r, ...] -> [sum n_r, ...] (rank order)."""
    if group is None:
        return t
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes + [1])
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def all_reduce_sum(t, group):
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(t, group=group)
    return t


def barrier(group):
    if group is not None:
        import torch.distributed as dist
        dist.barrier(group=group)
