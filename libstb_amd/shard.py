"""Discount-axis sharding for multi-GPU runs: tables for different discounts are
independent, so rank r owns a contiguous block of the sorted grid, fills and sweeps it locally, and
the only exchange is an all-gather of one scalar per discount (RCCL on GPUs, gloo in CPU tests)."""
from __future__ import annotations

import numpy as np


def my_slice(D_total: int, rank: int, world: int) -> slice:
    """contiguous block of the D_total grid points owned by `rank` (remainder spread over the
    first ranks, so any D_total works, not only multiples of world)"""
    base, extra = divmod(D_total, world)
    lo = rank * base + min(rank, extra)
    return slice(lo, lo + base + (1 if rank < extra else 0))


def counts(D_total: int, world: int):
    return [my_slice(D_total, r, world).stop - my_slice(D_total, r, world).start for r in range(world)]


def gather_scalars(local, D_total: int, dist=None, group=None, force_collective: bool = False, out=None):
    """all ranks end up with the D_total per-discount scalars in grid order.

    `local` is a 1-D torch tensor with this rank's values.  Equal shares use
    all_gather_into_tensor (one collective, 8 bytes per discount); ragged shares pad to the
    largest share first.  A single rank has nothing to exchange and gets a copy -- unless
    force_collective asks for the collective anyway (the one-GPU test of the RCCL path).
    `out`: a tensor of D_total values the result is written to (the collective writes it directly when the
    shares are equal: no temporary, no copy kernel behind it); returned."""
    import torch

    def deliver(res):
        if out is None:
            return res
        out.copy_(res)
        return out

    if dist is None or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return deliver(local) if out is not None else local.clone()
    world = dist.get_world_size(group)
    cnt = counts(D_total, world)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal mode (several ranks sharing one GPU, no RCCL): stage the 8-byte scalars through
        # the host; the production path below hands device tensors to RCCL directly
        return deliver(gather_scalars(local.cpu(), D_total, dist, group).to(local.device))
    if len(set(cnt)) == 1:
        res = out if (out is not None and out.is_contiguous() and out.numel() == D_total and out.dtype == local.dtype
                      and out.device == local.device) else torch.empty(D_total, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(res, local.contiguous(), group=group)
        return res if res is out else deliver(res)
    width = max(cnt)
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    buf = torch.empty(world * width, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return deliver(torch.cat([buf[r * width: r * width + cnt[r]] for r in range(world)]))


def max_over_ranks(seconds: float, device, dist=None) -> float:
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
