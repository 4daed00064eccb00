"""Site sharding across ranks: independent sites are partitioned contiguously,
each rank owns one engine with its share as `n_sites`, calibration needs NO communication, and the only
collective is the final all-gather of the per-site log-likelihoods (a few KB: latency-bound, one call)."""
import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous, balanced partition: the first n_total % world ranks get one extra site."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_sites(local, n_total: int, dist=None, device="cpu"):
    """All-gather per-site values (1-D float64 of this rank's shard) into the full [n_total] vector on every
    rank.  dist = torch.distributed (initialised: nccl = RCCL on ROCm, or gloo on CPU) or None (single rank)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert local.shape[0] == n_total
        return local.copy()
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    width = -(-n_total // world)  # equal-size slots: one fused collective, padded
    buf = torch.zeros(width, dtype=torch.float64, device=device)
    lo, hi = shard_range(n_total, rank, world)
    assert local.shape[0] == hi - lo
    buf[: hi - lo] = torch.from_numpy(local).to(device)
    out = torch.empty(world * width, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, width)
    full = np.empty(n_total)
    for r in range(world):
        a, b = shard_range(n_total, r, world)
        full[a:b] = out[r, : b - a]
    return full
