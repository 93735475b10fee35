"""Device-side minibatch sampler of the training loops: a fresh pseudo-random order of the training split per epoch,
never materialised -- batch rows are gathered by one HIP launch (csrc/shuffle.hip, ``sbi_amd_shuffled_gather``).

Plays the role of ``SubsetRandomSampler`` + the DataLoader's collation in sbi's loops
(sbi/inference/trainers/base.py:541-560).  The order of epoch ``e`` is the keyed permutation ``pi_{key(e)}`` of
``[0, n)``; ``key(e)`` is derived from one 62-bit seed per ``train()`` call (drawn from torch's global RNG on rank 0 and
broadcast), so all ranks walk the same orders with no per-epoch collective."""

from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from sbi_amd import _lib

_M64 = (1 << 64) - 1


def epoch_key(seed: int, epoch: int) -> int:
    """splitmix64 of (seed, epoch): the 64-bit Feistel key of one epoch's order."""
    z = (seed + 0x9E3779B97F4A7C15 * (epoch + 1)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def rank_window(lo: int, count: int, rank: int, world: int) -> Tuple[int, int]:
    """(offset, rows) of rank `rank`'s contiguous share of rows lo .. lo + count of an order: ceil(count / world) rows per
    rank, the last ranks possibly fewer or none -- the same split the index path makes (`my_slice` in the trainers)."""
    if world == 1:
        return lo, count
    per = (count + world - 1) // world
    a = min(rank * per, count)
    return lo + a, min((rank + 1) * per, count) - a


class ShuffledGather:
    """``batch(epoch, offset, count) -> (theta rows, x rows)`` of the epoch's order over ``base_idx`` (the training
    split's row numbers inside ``theta_all`` / ``x_all``); ``indices`` returns the source rows instead."""

    def __init__(self, theta_all: Tensor, x_all: Tensor, base_idx: Optional[Tensor], seed: int):
        self.dev = _lib.require_device(theta_all, x_all)
        if theta_all.dim() != 2 or x_all.dim() != 2 or theta_all.shape[0] != x_all.shape[0]:
            raise ValueError("ShuffledGather expects theta (N, D) and x (N, C)")
        if theta_all.dtype != torch.float32 or x_all.dtype != torch.float32:
            raise TypeError("ShuffledGather gathers fp32 rows")
        self.theta_all, self.x_all = theta_all.contiguous(), x_all.contiguous()
        self.base_idx = None if base_idx is None else base_idx.to(device=self.dev, dtype=torch.int64).contiguous()
        self.n = int(theta_all.shape[0] if base_idx is None else self.base_idx.numel())
        self.seed = int(seed)

    def _call(self, epoch: int, offset: int, count: int, a_out, b_out, idx_out) -> None:
        with torch.cuda.device(self.dev):
            rc = _lib.load().sbi_amd_shuffled_gather(
                _lib.ptr(self.theta_all), self.theta_all.shape[1], _lib.ptr(self.x_all), self.x_all.shape[1],
                None if self.base_idx is None else _lib.ptr(self.base_idx), self.n, epoch_key(self.seed, epoch),
                int(offset), int(count), None if a_out is None else _lib.ptr(a_out),
                None if b_out is None else _lib.ptr(b_out), None if idx_out is None else _lib.ptr(idx_out),
                _lib.current_stream(self.dev))
        _lib.check(rc, "shuffled_gather")

    def batch(self, epoch: int, offset: int, count: int) -> Tuple[Tensor, Tensor]:
        th = torch.empty(count, self.theta_all.shape[1], dtype=torch.float32, device=self.dev)
        xx = torch.empty(count, self.x_all.shape[1], dtype=torch.float32, device=self.dev)
        self._call(epoch, offset, count, th, xx, None)
        return th, xx

    def indices(self, epoch: int, offset: int, count: int) -> Tensor:
        idx = torch.empty(count, dtype=torch.int64, device=self.dev)
        self._call(epoch, offset, count, None, None, idx)
        return idx
