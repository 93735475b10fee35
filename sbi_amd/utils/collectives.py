"""The two collectives of the data-parallel training loops (reference loop
sbi/inference/trainers/base.py:1150-1193 has none -- sbi trains on one device): a SUM all-reduce of a flat fp32 buffer
and a rank-0 broadcast.

On the production path the process group is `nccl` (= RCCL over xGMI) and the device buffer goes to the collective as
it is, in stream order between the fused backward pass and the fused clip + Adam kernel.  A backend that cannot take a
device tensor (`gloo` in a build without device support -- the two-ranks-on-one-GPU test of tests/test_dp_two_rank_gpu.py
and `bench.py`'s SBI_AMD_BENCH_SHARE_GPU mode) gets the buffer staged through the host: same arithmetic (gloo sums in
rank order; with two ranks a + b is commutative, so replicas stay bit-identical), one synchronising copy each way.
"""

from __future__ import annotations

import torch
from torch import Tensor


_warned_staged = False


def device_capable(dist, group=None) -> bool:
    """Does the group's backend take ROCm device tensors?  Parsed from the backend string: a per-device map
    ("cpu:gloo,cuda:nccl") is judged by its `cuda` entry -- a missing or gloo `cuda` entry means staging -- a single name
    applies to every device ("nccl": yes, "gloo": no); anything unknown is trusted with the tensor as it is."""
    backend = str(dist.get_backend(group)).lower()
    if ":" in backend:
        entries = dict(e.split(":", 1) for e in backend.split(",") if ":" in e)
        return "cuda" in entries and entries["cuda"] != "gloo"
    return backend != "gloo"


def _direct(dist, t: Tensor, group=None) -> bool:
    """True when the group's backend reduces / broadcasts `t` where it lives.  Decided by capability: a device tensor
    is staged through the host only for a backend known to lack device support (pure `gloo`); `nccl`, the composite
    `cpu:gloo,cuda:nccl` a bare `init_process_group()` creates, and anything unknown get the tensor as it is.  The
    staged path costs two synchronising copies per call, so it announces itself once."""
    global _warned_staged
    if t.device.type == "cpu":
        return True
    if device_capable(dist, group):
        return True
    if not _warned_staged:
        import warnings

        warnings.warn("sbi_amd: the process group's backend is gloo: device buffers are staged through the host for every "
                      "collective (two synchronising copies per call); use backend 'nccl' (RCCL) for data-parallel "
                      "training on ROCm devices", stacklevel=3)
        _warned_staged = True
    return False


def all_reduce_sum(dist, t: Tensor, group=None) -> Tensor:
    """In-place SUM all-reduce of `t` over `group`; returns `t`."""
    if _direct(dist, t, group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t
    host = t.detach().to("cpu", copy=True)
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    t.copy_(host)
    return t


def broadcast_from_rank0(dist, t: Tensor, group=None) -> Tensor:
    """In-place broadcast of rank 0's `t`; returns `t`."""
    if _direct(dist, t, group):
        dist.broadcast(t, src=0, group=group)
        return t
    host = t.detach().to("cpu", copy=True)
    dist.broadcast(host, src=0, group=group)
    t.copy_(host)
    return t
