"""Multi-GPU layout: envs are independent, so a node runs one process per GPU, each owning a contiguous range of
env ids with its own replica of the read-only tables; nothing is exchanged on the data path.  The only collective is
the logging gather of finished-episode returns (RCCL all-gather over xGMI; `gloo` in the CPU tests).
The reference has no counterpart: its only parallelism is SB3's SubprocVecEnv, one OS process per env
(/root/reference/complete_pipeline.ipynb cell 13)."""
from __future__ import annotations

import os

import numpy as np

__all__ = ["dist_env", "shard_range", "gather_episode_stats"]


def dist_env() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched distributed."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(total_envs: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous [lo, hi) env-id range of `rank`; the first total_envs % world_size ranks hold one extra env."""
    base, extra = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_episode_stats(returns, lengths, group=None, equal_shards: bool = False):
    """All-gather per-rank `returns` f64[E_r] and `lengths` i32[E_r] into global arrays ordered by env id.
    Accepts torch tensors on the collective's device (GPU for nccl/RCCL, CPU for gloo) or NumPy arrays (CPU).
    Ranks may hold different numbers of envs (padded to the maximum for the collective, trimmed afterwards);
    `equal_shards=True` promises equal counts and does the whole exchange in ONE collective (returns and lengths packed
    into one float64 buffer: episode lengths are exact in float64)."""
    import torch
    import torch.distributed as dist

    as_numpy = isinstance(returns, np.ndarray)
    r = torch.as_tensor(returns, dtype=torch.float64)
    n = torch.as_tensor(lengths, dtype=torch.int32)
    if not (dist.is_available() and dist.is_initialized()):
        return (r.numpy(), n.numpy()) if as_numpy else (r, n)
    world = dist.get_world_size(group)
    if equal_shards:
        m = r.numel()
        packed = torch.cat([r, n.to(torch.float64)])
        out = torch.empty(world * 2 * m, dtype=torch.float64, device=r.device)
        dist.all_gather_into_tensor(out, packed, group=group)
        out = out.view(world, 2, m)
        r_all = out[:, 0, :].reshape(-1)
        n_all = out[:, 1, :].reshape(-1).to(torch.int32)
        return (r_all.cpu().numpy(), n_all.cpu().numpy()) if as_numpy else (r_all, n_all)
    count = torch.tensor([r.numel()], dtype=torch.int64, device=r.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    sizes = [int(c.item()) for c in counts]
    m = max(sizes)
    rp = torch.zeros(m, dtype=torch.float64, device=r.device)
    npad = torch.zeros(m, dtype=torch.int32, device=r.device)
    rp[: r.numel()] = r
    npad[: n.numel()] = n
    rg = [torch.empty_like(rp) for _ in range(world)]
    ng = [torch.empty_like(npad) for _ in range(world)]
    dist.all_gather(rg, rp, group=group)
    dist.all_gather(ng, npad, group=group)
    r_all = torch.cat([x[:k] for x, k in zip(rg, sizes)])
    n_all = torch.cat([x[:k] for x, k in zip(ng, sizes)])
    return (r_all.cpu().numpy(), n_all.cpu().numpy()) if as_numpy else (r_all, n_all)
