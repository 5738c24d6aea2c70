"""Data-parallel serving helpers: one process per GPU, full weight replica each, requests sharded by rank.

The VisualCLA path has no cross-sample reduction anywhere, so the only collective is a single
all-gather of the generated token ids (KBs; latency-bound over xGMI).  `torch.distributed` backend "nccl" is RCCL on
ROCm; the same code runs on "gloo" for the CPU tests."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n requests: the first (n % world) ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_tokens(tokens: torch.Tensor, n_total: Optional[int] = None, n_cols: Optional[int] = None, pad_id: int = 0,
                  force_collective: bool = False) -> torch.Tensor:
    """All-gather [b_rank, n] token ids from every rank into [n_total, n_cols] (rank order = request order) with ONE collective.
    The shard sizes follow from `shard_range(n_total, rank, world)` and every shard is padded to ceil(n_total / world) rows x
    `n_cols` columns (= max_new_tokens; early EOS leaves pad_id behind), so nothing has to be exchanged up front and nothing
    synchronises the host.  Without n_total / n_cols (ragged callers) the shapes are exchanged first: two collectives.
    A world of one returns its own shard without touching the backend unless `force_collective` is set (the 1-GPU check that RCCL
    initialises and moves int64 device tensors on this box: tests/test_gpu_distributed.py, bench.py --force-collective)."""
    if not (dist.is_available() and dist.is_initialized()):
        return tokens
    if dist.get_world_size() == 1 and not force_collective:
        return tokens
    world = dist.get_world_size()
    dev = tokens.device
    if n_total is not None and n_cols is not None:
        bmax = (n_total + world - 1) // world
        if tokens.shape[0] > bmax or tokens.shape[1] > n_cols:
            raise ValueError(f"shard {tuple(tokens.shape)} exceeds the padded shard shape ({bmax}, {n_cols})")
        buf = torch.full((bmax, n_cols), pad_id, dtype=tokens.dtype, device=dev)
        buf[: tokens.shape[0], : tokens.shape[1]] = tokens
        out = torch.empty(world * bmax, n_cols, dtype=tokens.dtype, device=dev)
        dist.all_gather_into_tensor(out, buf)
        if n_total == world * bmax:
            return out
        sizes = [shard_range(n_total, r, world) for r in range(world)]
        return torch.cat([out.view(world, bmax, n_cols)[r, : hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
    shape = torch.tensor(list(tokens.shape), dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    bmax = int(max(int(s[0]) for s in shapes))
    nmax = int(max(int(s[1]) for s in shapes))
    buf = torch.full((bmax, nmax), pad_id, dtype=tokens.dtype, device=dev)
    buf[: tokens.shape[0], : tokens.shape[1]] = tokens
    out = torch.empty(world * bmax, nmax, dtype=tokens.dtype, device=dev)
    dist.all_gather_into_tensor(out, buf)
    parts = [out.view(world, bmax, nmax)[r, : int(shapes[r][0])] for r in range(world)]
    return torch.cat(parts, dim=0)
