"""Device-resident NPE training step on the fused HIP kernels.

One ``FusedTrainStep.step(theta, x)`` replaces the body of sbi's hot loop
(sbi/inference/trainers/base.py:1173-1187):

    optimizer.zero_grad(); losses = net.loss(theta, x); loss = losses.mean()
    loss.backward(); clip_grad_norm_(net.parameters(), 5.0); optimizer.step()

with: weight re-pack -> fused loss forward+backward (flat gradient of the batch
mean) -> [data parallel: ONE all-reduce of the flat 98 025-float gradient over
RCCL] -> fused global-norm clip + Adam.  No host synchronisation happens inside
a step; per-row losses stay on the device.
"""

from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from sbi_amd import _lib
from sbi_amd.neural_nets.estimators.nsf_flow import NSFFlow, train_backward, train_forward
from sbi_amd.utils.collectives import all_reduce_sum


class FusedTrainStep:
    def __init__(self, estimator: NSFFlow, lr: float = 5e-4, clip_max_norm: Optional[float] = 5.0,
                 betas=(0.9, 0.999), eps: float = 1e-8, distributed: bool = False, process_group=None,
                 native_allreduce: bool = False):
        self.est = estimator
        self.net = estimator.net
        p = self.net.flat_params
        _lib.require_device(p)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.clip = float(clip_max_norm) if clip_max_norm is not None else 0.0
        self.exp_avg = torch.zeros_like(p.data)
        self.exp_avg_sq = torch.zeros_like(p.data)
        self.grad = torch.zeros_like(p.data)
        self.scratch = torch.zeros(256, dtype=torch.float32, device=p.device)
        self.workspace: Optional[Tensor] = None
        self.step_count = 0
        self.distributed = distributed
        self.group = process_group
        if distributed:
            import torch.distributed as dist

            self.dist = dist
            self.world = dist.get_world_size(process_group)
        else:
            self.world = 1
        # the gradient all-reduce through the library's own RCCL entry point (sbi_amd_allreduce_flat) instead of
        # torch.distributed.all_reduce: what a C host binds; the process group only ships the communicator id once
        self._native_allreduce = None
        if distributed and native_allreduce:
            from sbi_amd.utils.collectives import NativeAllReduce

            self._native_allreduce = NativeAllReduce(self.dist, p.device, process_group)

    # -- state for resume_training / best-weights bookkeeping -----------------------
    def state_dict(self):
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "step": self.step_count}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])

    # -- device snapshots for the software-pipelined epoch loop of NPE.train() -------------------
    def snapshot(self) -> dict:
        return {"params": self.net.flat_params.data.clone(), "exp_avg": self.exp_avg.clone(),
                "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.step_count}

    def snapshot_into(self, snap: Optional[dict]) -> dict:
        """`snapshot()` into the buffers of an earlier snapshot (three plain copies, no allocation); `None` allocates.
        The epoch loop keeps a small ring of them.  (A multi-tensor `_foreach_copy_` is one launch instead of three but
        costs MORE host time, and the one-step-per-epoch loop is host-bound on slow boxes: measured, not kept.)"""
        if snap is None:
            return self.snapshot()
        snap["params"].copy_(self.net.flat_params.data)
        snap["exp_avg"].copy_(self.exp_avg)
        snap["exp_avg_sq"].copy_(self.exp_avg_sq)
        snap["step"] = self.step_count
        return snap

    def restore_optimizer(self, snap: dict) -> None:
        self.exp_avg.copy_(snap["exp_avg"])
        self.exp_avg_sq.copy_(snap["exp_avg_sq"])
        self.step_count = int(snap["step"])

    # images (bit 0 throughput, bit 1 cooperative) the step's table-driven re-pack refreshes IN ADDITION to the one the
    # training batches read: a loop whose validation batches take the other kernel family and that has only a step or
    # two per epoch saves the separate pack launch per epoch (NPE.train sets it; 0: only the training image)
    tail_extra_images = 0
    def restore(self, snap: dict) -> None:
        """Weights AND optimizer state of a snapshot.  The write goes behind autograd's back (`.data`), so the tensor
        version the packed-weight cache is keyed on does not move: drop the cache, or log_prob / sample after a
        recovered NaN epoch would keep running on the poisoned image."""
        self.net.flat_params.data.copy_(snap["params"])
        self.restore_optimizer(snap)
        self.net.__dict__.pop("_packed_cache", None)

    def _workspace(self, n: int) -> Tensor:
        need = self.net.train_workspace_floats(n)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(int(need), dtype=torch.float32, device=self.net.flat_params.device)
        return self.workspace

    def _embedded(self, x: Tensor) -> Tensor:
        """The kernels consume the EMBEDDED condition: a frozen / parameter-free embedding net (and the
        `Standardize` layer build_nsf puts in front of it, flow.py:1395-1416) is applied here, without a graph.
        Estimators with a trainable embedding never reach the fused step (NPE.train takes the autograd path)."""
        emb = getattr(self.est, "_embedding_net", None)
        if emb is None:
            return x
        if any(p.requires_grad for p in emb.parameters()):
            raise RuntimeError("FusedTrainStep cannot train an embedding net (its optimizer owns the flow's flat "
                               "parameter buffer only); use the autograd path of NPE.train().")
        with torch.no_grad():
            return self.est._embed(x).contiguous().float()

    @torch.no_grad()
    def loss_and_grad(self, theta: Tensor, x: Tensor, global_batch: Optional[int] = None) -> Tensor:
        """Per-row losses (device tensor); leaves d(mean loss)/d(params) in ``self.grad``
        (summed over ranks when distributed).  Whoever changes ``self.grad`` between this call and ``apply()`` calls
        ``grad_modified()`` (the clip's norm otherwise comes from partial sums of the pass's own reduction)."""
        n = theta.shape[0]
        self._last_rows = n
        gb = global_batch if global_batch is not None else n * self.world
        x = self._embedded(x)
        losses, _ = self.net.train_pass(theta, x, None, 1.0 / gb, self.grad, workspace=self._workspace(n))
        if self._native_allreduce is not None:
            self._native_allreduce(self.grad)
        elif self.distributed:
            all_reduce_sum(self.dist, self.grad, self.group)
        self._mark_grad_from_pass()
        if self._native_allreduce is not None and self.world > 1:
            self._grad_from_pass = False      # (written through a raw pointer: the pass's |grad|^2 partials are stale)
        return losses

    def _mark_grad_from_pass(self) -> None:
        """self.grad is exactly what the pass's reduction kernel wrote: `apply()` may take |grad|^2 from the partial sums
        that kernel left behind.  The tensor's version counter is remembered, so ANY later in-place torch operation on
        `self.grad` (scaling, accumulation, ...) silently withdraws that permission -- `grad_modified()` is only needed
        by callers that write through a raw pointer."""
        self._grad_from_pass = True
        self._grad_version = self.grad._version

    @torch.no_grad()
    def atomic_loss_and_grad(self, theta: Tensor, x: Tensor, masks: Tensor, prior, num_atoms: int,
                             use_combined_loss: bool = False, global_batch: Optional[int] = None,
                             choices: Optional[Tensor] = None) -> Tensor:
        """Multi-round NPE-C: per-row atomic proposal-posterior loss (npe_c.py:356-440); leaves
        d(sum loss / global_batch)/d(params) in ``self.grad``.  One forward pass over the A*B (theta, x)
        pairs (atoms-major: row r is conditioned on x[r % B], the context is never repeated), the softmax
        weights d loss / d log q on the device, one backward pass on the stash of that forward."""
        from sbi_amd.inference.trainers.npe.atomic import build_atoms, clamp_num_atoms, sample_contrasting_indices

        if not getattr(self.net, "supports_atomic", False):
            raise NotImplementedError(f"{type(self.net).__name__} has no split forward / backward training pass; "
                                      "multi-round training of this estimator takes NPE.train()'s autograd path")
        B = theta.shape[0]
        x = self._embedded(x)
        A = clamp_num_atoms(num_atoms, B)
        self._last_rows = A * B
        gb = global_batch if global_batch is not None else B * self.world
        ws = self._workspace(A * B)
        fused_host = (theta.is_cuda and theta.dtype == torch.float32 and theta.ndim == 2 and A - 1 <= 63
                      and (choices is None or choices.is_cuda))
        if fused_host:
            # the arithmetic around the batched log_prob as two launches (csrc/atomic.hip) instead of ~80 eager tensor
            # operations: contrasting rows + atom tensor, then log q~ and the softmax weights of the backward pass
            lib = _lib.load()
            dev = theta.device
            th = theta.contiguous()
            flat = torch.empty(A * B, th.shape[1], dtype=torch.float32, device=dev)
            seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item()) if choices is None else 0
            ch = None if choices is None else choices.to(torch.int64).contiguous()
            with torch.cuda.device(dev):
                rc = lib.sbi_amd_atomic_atoms(_lib.ptr(th), B, A, th.shape[1], seed, _lib.ptr(ch), None, _lib.ptr(flat),
                                              _lib.current_stream(dev))
            _lib.check(rc, "atomic_atoms")
            lp = train_forward(self.net, flat, x, ws).reshape(-1).contiguous()
            lprior = prior.log_prob(flat).reshape(-1).to(torch.float32).contiguous()
            m = masks.reshape(-1).to(torch.float32).contiguous() if use_combined_loss else None
            lpp = torch.empty(B, dtype=torch.float32, device=dev)
            w = torch.empty(A * B, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.sbi_amd_atomic_weights(_lib.ptr(lp), _lib.ptr(lprior), _lib.ptr(m), B, A, 1.0 / gb, _lib.ptr(lpp),
                                                _lib.ptr(w), _lib.current_stream(dev))
            _lib.check(rc, "atomic_weights")
            train_backward(self.net, x, A * B, w, self.grad, ws)
        else:
            if choices is None:
                choices = sample_contrasting_indices(B, A, theta.device)
            flat = build_atoms(theta, choices).reshape(A * B, -1).contiguous()
            lp = train_forward(self.net, flat, x, ws).reshape(A, B)
            lprior = prior.log_prob(flat).reshape(A, B)
            un = lp - lprior
            lse = torch.logsumexp(un, dim=0)
            lpp = un[0] - lse
            w = -torch.exp(un - lse)          # d lpp_b / d log q[a, b] = delta_{a0} - softmax_a
            w[0] += 1.0
            if use_combined_loss:             # + masks * log q(theta_b | x_b): the same values as atom 0
                m = masks.reshape(-1).to(lp.dtype)
                lpp = m * lp[0] + lpp
                w[0] += m
            train_backward(self.net, x, A * B, (w / gb).reshape(-1).contiguous(), self.grad, ws)
        if self.distributed:
            all_reduce_sum(self.dist, self.grad, self.group)
        self._mark_grad_from_pass()
        return -lpp

    def atomic_step(self, theta: Tensor, x: Tensor, masks: Tensor, prior, num_atoms: int,
                    use_combined_loss: bool = False, global_batch: Optional[int] = None) -> Tensor:
        losses = self.atomic_loss_and_grad(theta, x, masks, prior, num_atoms, use_combined_loss, global_batch)
        self.apply()
        return losses

    # -- step tail (csrc/step_tail.hip): the re-pack of the image the next step reads, from a gather table built once
    def _tail(self):
        """(images mask, packed buffer, gather table) for the table-driven re-pack, or None: not an NSF net, rows of the
        last pass unknown, switched off (SBI_AMD_FUSED_TAIL=0), or the table could not be built for this shape."""
        import os

        from sbi_amd.neural_nets.estimators.nsf_flow import NSFNet, packed_weights

        rows = getattr(self, "_last_rows", None)
        if type(self.net) is not NSFNet or rows is None or os.environ.get("SBI_AMD_FUSED_TAIL", "1") == "0":
            return None
        lib = _lib.load()
        cfg = self.net.hyper.c_config()
        kind = lib.sbi_amd_nsf_image_kind(cfg, int(rows), 1)
        if kind < 0:
            return None
        mask = (2 if kind == 1 else 1) | int(getattr(self, "tail_extra_images", 0))
        packed = packed_weights(self.net, rows=int(rows), training=True)      # (build_step_map packs every image of `mask`)
        maps = self.__dict__.setdefault("_step_maps", {})
        key = (mask, packed.data_ptr(), self.net.flat_params.data_ptr())
        for stale in [k for k in maps if k[1:] != key[1:]]:      # a moved / re-allocated buffer drops its tables
            self._drop_step_map(maps.pop(stale))
        ent = maps.get(key)                                       # (at most one table per image: batch sizes on both
        if ent is None:                                           #  sides of the kernel families' threshold keep two)
            dev = packed.device
            n_map = lib.sbi_amd_nsf_step_map_ints(cfg)
            n_ws = lib.sbi_amd_nsf_step_map_workspace_floats(cfg)
            if n_map < 0 or n_ws < 0:
                ent = False
            else:
                mp = torch.zeros(int(n_map), dtype=torch.int32, device=dev)
                ws = torch.empty(int(n_ws), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    rc = lib.sbi_amd_nsf_build_step_map(cfg, mask, _lib.ptr(self.net.flat_params.data),
                                                        _lib.ptr(packed), _lib.ptr(mp), _lib.ptr(ws),
                                                        _lib.current_stream(dev))
                if rc == _lib.E_UNSUPPORTED:
                    import warnings

                    warnings.warn("sbi_amd: the re-pack table could not be built for this network; every step "
                                  "re-packs with sbi_amd_nsf_pack_images", stacklevel=2)
                    ent = False
                else:
                    _lib.check(rc, "nsf_build_step_map")
                    ent = mp
                    # the library keeps a registry of built tables by address: release the entry when the tensor goes
                    import weakref

                    weakref.finalize(mp, lib.sbi_amd_nsf_release_step_map, mp.data_ptr())
                del ws
            maps[key] = ent
        if ent is False:
            return None
        return mask, packed, ent

    @staticmethod
    def _drop_step_map(ent) -> None:
        if isinstance(ent, torch.Tensor):
            _lib.load().sbi_amd_nsf_release_step_map(ent.data_ptr())

    def grad_modified(self) -> None:
        """Tell the stepper that `self.grad` is no longer what the last training pass wrote (the caller scaled it,
        accumulated into it, ...): the next `apply()` takes |grad|^2 from the gradient itself instead of from the
        partial sums the pass's reduction kernel left behind."""
        self._grad_from_pass = False

    def _norm_parts(self):
        """(pointer, count) of the partial sums of squares of `self.grad` that the last NSF training pass's gradient
        reduction left in the workspace (csrc/nsf_train.hip, a rider of the reduction kernel), or None: not an NSF net,
        the pass leaves none (generic training pass), switched off (SBI_AMD_NORM_RIDER=0) -- or MORE THAN ONE rank: the
        all-reduce changed the gradient, the norm has to be taken from the reduced one (`grad_sqnorm_partials`).  With
        one rank the all-reduce is the identity, so the 1-rank RCCL run and the group-less run take the same path."""
        import ctypes
        import os

        from sbi_amd.neural_nets.estimators.nsf_flow import NSFNet

        rows = getattr(self, "_last_rows", None)
        if (type(self.net) is not NSFNet or rows is None or self.world != 1 or self.workspace is None
                or not getattr(self, "_grad_from_pass", False) or self.grad._version != getattr(self, "_grad_version", -1)
                or os.environ.get("SBI_AMD_NORM_RIDER", "1") == "0"):
            return None
        n_parts = ctypes.c_int64(0)
        ptr = _lib.load().sbi_amd_nsf_train_sqnorm_parts(self.net.hyper.c_config(), int(rows), _lib.ptr(self.workspace),
                                                         ctypes.byref(n_parts))
        if not ptr or n_parts.value < 1:
            return None
        return ptr, int(n_parts.value)

    @torch.no_grad()
    def apply(self) -> None:
        """Fused clip_grad_norm_ + Adam on the flat buffers, then the table-driven re-pack of the image the next
        pass reads (csrc/step_tail.hip) when a table exists for this network."""
        lib = _lib.load()
        p = self.net.flat_params
        dev = p.device
        tail = self._tail()
        self.step_count += 1
        parts = self._norm_parts()
        self._grad_from_pass = False         # (whoever changes self.grad by hand and calls apply() again gets the norm kernel)
        with torch.cuda.device(dev):
            if parts is not None:      # |grad|^2 as partial sums the gradient reduction left in the workspace
                rc = lib.sbi_amd_adam_clip_step_parts(
                    _lib.ptr(p.data), _lib.ptr(self.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                    p.numel(), self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.clip,
                    parts[0], parts[1], _lib.ptr(self.scratch), _lib.current_stream(dev),
                )
            else:
                rc = lib.sbi_amd_adam_clip_step(
                    _lib.ptr(p.data), _lib.ptr(self.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                    p.numel(), self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.clip,
                    _lib.ptr(self.scratch), _lib.current_stream(dev),
                )
        _lib.check(rc, "adam_clip_step")
        if tail is not None:
            mask, packed, mp = tail
            with torch.cuda.device(dev):
                rc = lib.sbi_amd_nsf_table_pack(self.net.hyper.c_config(), _lib.ptr(p.data), _lib.ptr(packed),
                                                _lib.ptr(mp), _lib.current_stream(dev))
            if rc == _lib.E_BADARG:
                # the library no longer knows this table (released, or built for another configuration): the optimizer
                # step has already happened, so do not raise here -- forget the table (it is rebuilt on the next step)
                # and let packed_weights re-pack from the new parameters with the full pack kernels
                for k in [k for k, v in self.__dict__.get("_step_maps", {}).items() if v is mp]:
                    self._drop_step_map(self._step_maps.pop(k))
                cache = self.net.__dict__.get("_packed_cache")
                if cache is not None:
                    self.net.__dict__["_packed_cache"] = (None, cache[1])
                return
            _lib.check(rc, "nsf_table_pack")
            # the image named by `mask` holds the new parameters; every other image (and the explicit LU inverses) is
            # stale: packed_weights re-packs those on demand
            self.net.__dict__["_packed_cache"] = ((p.data_ptr(), p._version, str(p.device)), packed)
            self.net.__dict__["_packed_images"] = mask
            return
        # parameters changed in place behind autograd's back: invalidate the packed image (the buffer itself is kept:
        # its alignment gaps are zero and stay zero, a fresh torch.zeros per step is a launch for nothing)
        cache = self.net.__dict__.get("_packed_cache")
        if cache is not None:
            self.net.__dict__["_packed_cache"] = (None, cache[1])

    def step(self, theta: Tensor, x: Tensor, global_batch: Optional[int] = None) -> Tensor:
        losses = self.loss_and_grad(theta, x, global_batch)
        self.apply()
        return losses

    def grad_norm(self) -> Tensor:
        """Pre-clip gradient norm of the last ``apply`` (device scalar)."""
        return self.scratch[0]


class FusedFMPEStep(FusedTrainStep):
    """The same device-resident step for the flow-matching estimator (FMPE): draw t ~ U[0, 1] and
    theta_1 ~ N(0, I) on the device, fused CFM loss forward + backward (csrc/fmpe.hip), [one all-reduce of the
    flat gradient], fused clip + Adam.  Replaces the loop body of VectorFieldTrainer's training epoch
    (sbi/inference/trainers/vfpe/base_vf_inference.py:588-625 -> trainers/base.py:1160-1190)."""

    def _workspace(self, n: int) -> Tensor:
        from sbi_amd.neural_nets.estimators.flowmatching_estimator import train_workspace

        self.workspace = train_workspace(self.net, n, self.net.flat_params.device, self.workspace)
        return self.workspace

    @torch.no_grad()
    def loss_and_grad(self, theta: Tensor, x: Tensor, global_batch: Optional[int] = None,
                      times: Optional[Tensor] = None, noise: Optional[Tensor] = None,
                      row_weight: Optional[Tensor] = None) -> Tensor:
        from sbi_amd.neural_nets.estimators.flowmatching_estimator import loss_fwd_bwd as fm_loss_fwd_bwd

        n = theta.shape[0]
        gb = global_batch if global_batch is not None else n * self.world
        if times is None:
            times = torch.rand(n, device=theta.device, dtype=torch.float32)
        if noise is None:
            noise = torch.randn_like(theta)
        if row_weight is not None:
            row_weight = (row_weight / gb).contiguous()
        losses = fm_loss_fwd_bwd(self.net, theta, x, times, noise, row_weight, 1.0 / gb, self.grad,
                                 workspace=self._workspace(n))
        if self.distributed:
            all_reduce_sum(self.dist, self.grad, self.group)
        return losses

    def atomic_loss_and_grad(self, *a, **k):
        raise NotImplementedError("multi-round FMPE with arbitrary proposals is not implemented (as in sbi)")
