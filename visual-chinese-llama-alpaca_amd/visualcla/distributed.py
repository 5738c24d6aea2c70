"""Data-parallel serving helpers: one process per GPU, full weight replica each, requests sharded by rank.

The VisualCLA path has no cross-sample reduction anywhere, so the only collective is a single
all-gather of the generated token ids (KBs; latency-bound over xGMI).  `torch.distributed` backend "nccl" is RCCL on
ROCm; the same code runs on "gloo" for the CPU tests."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n requests: the first (n % world) ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_tokens(tokens: torch.Tensor, pad_id: int = 0) -> torch.Tensor:
    """All-gather [b_rank, n] token ids from every rank into [sum b_rank, n_max] (rank order = request order).
    Shards may be ragged in both dims (uneven split, early EOS): sizes are exchanged first, payloads are padded."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tokens
    world = dist.get_world_size()
    dev = tokens.device
    shape = torch.tensor(list(tokens.shape), dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    bmax = int(max(int(s[0]) for s in shapes))
    nmax = int(max(int(s[1]) for s in shapes))
    buf = torch.full((bmax, nmax), pad_id, dtype=tokens.dtype, device=dev)
    buf[: tokens.shape[0], : tokens.shape[1]] = tokens
    out = torch.empty(world * bmax, nmax, dtype=tokens.dtype, device=dev)
    dist.all_gather_into_tensor(out, buf) if dev.type == "cuda" else dist.all_gather(list(out.view(world, bmax, nmax).unbind(0)), buf)
    parts = [out.view(world, bmax, nmax)[r, : int(shapes[r][0])] for r in range(world)]
    return torch.cat(parts, dim=0)
