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


class NativeAllReduce:
    """The step's gradient all-reduce through the kernel library's own entry point (`sbi_amd_allreduce_flat`: RCCL
    resolved by the library, include/sbi_amd_nsf.h) instead of `torch.distributed.all_reduce` -- the path a C / C++ host
    takes.  `dist` (any initialised torch.distributed group, e.g. gloo) is only used ONCE, to ship rank 0's 128-byte
    RCCL id to the other ranks.  Call it on a contiguous fp32 device tensor; the reduction is enqueued on the current
    stream of the tensor's device."""

    def __init__(self, dist, device, group=None):
        import ctypes

        from sbi_amd import _lib

        self._lib = _lib
        lib = _lib.load()
        self.device = torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        nbytes = int(lib.sbi_amd_rccl_unique_id_bytes())
        box = [None]
        if self.rank == 0:
            buf = (ctypes.c_char * nbytes)()
            _lib.check(lib.sbi_amd_rccl_unique_id(buf), "rccl_unique_id")
            box[0] = bytes(buf)
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.sbi_amd_rccl_comm_init(ctypes.byref(comm), self.world, self.rank, box[0]), "rccl_comm_init")
        self._comm = comm

    def __call__(self, t: Tensor) -> Tensor:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            raise ValueError("NativeAllReduce takes a contiguous float32 tensor on the communicator's device")
        with torch.cuda.device(self.device):
            rc = self._lib.load().sbi_amd_allreduce_flat(self._comm, self._lib.ptr(t), t.numel(),
                                                         self._lib.current_stream(self.device))
        self._lib.check(rc, "allreduce_flat")
        return t

    def close(self) -> None:
        if getattr(self, "_comm", None):
            self._lib.load().sbi_amd_rccl_comm_destroy(self._comm)
            self._comm = None
