"""`maf_rqs` density estimator on the MI355X HIP kernels.

``MAFRQSFlow`` is the drop-in for sbi's ``NFlowsFlow(build_maf_rqs(...))``
(sbi/neural_nets/net_builders/flow.py:212-330, estimators/nflows_flow.py:14-151): a masked autoregressive flow whose
elementwise transforms are rational-quadratic splines with linear tails, T x [MADE-conditioned spline on all D dims,
RandomPermutation].  The estimator surface (shapes, broadcasting, RNG order of ``sample``) is inherited from
``NSFFlow``; the arithmetic runs in ``libsbi_amd_nsf.so`` through include/sbi_amd_maf.h.  No PyTorch / CPU fallback.

Parameters live in ONE flat fp32 ``nn.Parameter`` in nflows' order (initial_layer, context_layer, blocks.b.linear,
final_layer per transform); the MADE degree masks are static and are multiplied into the packed weight image (and
into the weight gradients) by the kernels, so masked entries of the flat buffer never matter.  The permutations are
an int32 buffer ``perms`` (T, D), drawn with ``torch.randperm`` in nflows' construction order.
"""

from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from sbi_amd import _lib
from sbi_amd.neural_nets.estimators.nsf_flow import NSFFlow


@dataclass(frozen=True)
class MAFHyper:
    """Hyper-parameters ``build_maf_rqs`` bakes into the flow (flow.py:212-235)."""

    D: int
    C: int
    hidden_features: int = 50
    num_transforms: int = 5
    num_bins: int = 10
    num_blocks: int = 2
    tail_bound: float = 3.0
    min_bin_width: float = 1e-3
    min_bin_height: float = 1e-3
    min_derivative: float = 1e-3
    scale_by_sqrt_hidden: bool = False   # nflows' MADE exposes no `hidden_features`: logits are used unscaled

    def c_config(self) -> _lib.MAFConfigC:
        return _lib.MAFConfigC(self.D, self.C, self.hidden_features, self.num_bins, self.num_transforms,
                               self.num_blocks, self.tail_bound, self.min_bin_width, self.min_bin_height,
                               self.min_derivative, int(self.scale_by_sqrt_hidden), 0)

    def layer_entries(self) -> List[Tuple[str, Tuple[int, ...], int]]:
        """(nflows sub-key, shape, mask kind) in flat order for one transform (kinds as csrc/maf_kernel.h maf_mask)."""
        H, D, C, P = self.hidden_features, self.D, self.C, 3 * self.num_bins - 1
        pre = "autoregressive_net."
        out = [(pre + "initial_layer.weight", (H, D), 0), (pre + "initial_layer.bias", (H,), -1),
               (pre + "context_layer.weight", (H, C), 1), (pre + "context_layer.bias", (H,), -1)]
        for b in range(self.num_blocks):
            out += [(pre + f"blocks.{b}.linear.weight", (H, H), 2), (pre + f"blocks.{b}.linear.bias", (H,), -1)]
        out += [(pre + "final_layer.weight", (D * P, H), 3), (pre + "final_layer.bias", (D * P,), -1)]
        return out

    def layer_params(self) -> int:
        return sum(int(np.prod(s)) for _, s, _ in self.layer_entries())

    def param_count(self) -> int:
        return self.num_transforms * self.layer_params()

    # -- MADE degrees / masks (nflows transforms/made.py, random_mask=False) -----------------------
    def hidden_degrees(self) -> Tensor:
        mx, mn = max(1, self.D - 1), min(1, self.D - 1)
        return torch.arange(self.hidden_features) % mx + mn

    def mask(self, kind: int) -> Optional[Tensor]:
        D, P = self.D, 3 * self.num_bins - 1
        hd = self.hidden_degrees()
        if kind == 0:
            return (hd[:, None] >= torch.arange(1, D + 1)[None, :]).float()
        if kind == 2:
            return (hd[:, None] >= hd[None, :]).float()
        if kind == 3:
            od = torch.repeat_interleave(torch.arange(1, D + 1), P)
            return (od[:, None] > hd[None, :]).float()
        return None


class MAFNet(nn.Module):
    """Parameter / buffer holder in the role of nflows' ``Flow`` for maf_rqs."""

    supports_atomic = False    # one-call training pass only (no split forward / backward on a shared stash)

    def __init__(self, hyper: MAFHyper, zstats: Tensor, z_score_theta: bool, z_score_x: bool,
                 dtype: torch.dtype = torch.float32):
        super().__init__()
        self.hyper = hyper
        self.z_score_theta = z_score_theta
        self.z_score_x = z_score_x
        self.flat_params = nn.Parameter(torch.zeros(hyper.param_count(), dtype=torch.float32))
        self.register_buffer("zstats", zstats.to(torch.float32).contiguous())
        self.register_buffer("perms", torch.zeros(hyper.num_transforms, hyper.D, dtype=torch.int32))
        self.register_buffer("_log_z", torch.tensor(0.5 * hyper.D * math.log(2 * math.pi),
                                                    dtype=torch.float64).to(dtype), persistent=False)
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self) -> None:
        """nflows-equivalent initialisation in nflows' construction order (per transform: MaskedLinear initial,
        nn.Linear context, block linears, final MaskedLinear -- each the default nn.Linear init -- then
        RandomPermutation's torch.randperm), drawing from torch's global generator."""
        h = self.hyper
        H, D, C, P = h.hidden_features, h.D, h.C, 3 * h.num_bins - 1
        chunks: List[Tensor] = []
        for t in range(h.num_transforms):
            mods = [nn.Linear(D, H), nn.Linear(C, H)] + [nn.Linear(H, H) for _ in range(h.num_blocks)] + \
                   [nn.Linear(H, D * P)]
            for m in mods:
                chunks += [m.weight.detach().reshape(-1), m.bias.detach().reshape(-1)]
            self.perms[t] = torch.randperm(D).to(torch.int32)
        flat = torch.cat(chunks)
        assert flat.numel() == self.flat_params.numel()
        self.flat_params.copy_(flat)

    # -- nflows state_dict exchange ------------------------------------------------------------------
    def _slices(self):
        h = self.hyper
        first = 1 if self.z_score_theta else 0
        off = 0
        for t in range(h.num_transforms):
            pre = f"_transform._transforms.{first + 2 * t}."
            for key, shape, _kind in h.layer_entries():
                n = int(np.prod(shape))
                yield pre + key, off, n, shape
                off += n

    def nflows_state_dict(self, prefix: str = "net.") -> "OrderedDict[str, Tensor]":
        h = self.hyper
        sd: "OrderedDict[str, Tensor]" = OrderedDict()
        flat = self.flat_params.detach()
        first = 1 if self.z_score_theta else 0
        if self.z_score_theta:
            sd[prefix + "_transform._transforms.0._shift"] = self.zstats[: h.D].clone()
            sd[prefix + "_transform._transforms.0._scale"] = self.zstats[h.D : 2 * h.D].clone()
        for key, off, n, shape in self._slices():
            sd[prefix + key] = flat[off : off + n].reshape(shape).clone()
        # nflows' MaskedLinear registers its `mask` and `degrees` as buffers (made.py): a strict load into a real
        # nflows Flow needs them (round-2 advisor finding)
        for t in range(h.num_transforms):
            for key, mask, degrees in self._mask_buffers(t):
                sd[prefix + key + "mask"] = mask
                sd[prefix + key + "degrees"] = degrees
        for t in range(h.num_transforms):
            sd[prefix + f"_transform._transforms.{first + 2 * t + 1}._permutation"] = self.perms[t].to(torch.int64)
        if self.z_score_x:
            sd[prefix + "_embedding_net.0._mean"] = self.zstats[2 * h.D : 2 * h.D + h.C].clone()
            sd[prefix + "_embedding_net.0._std"] = self.zstats[2 * h.D + h.C :].clone()
        return sd

    def _mask_buffers(self, t: int):
        """(module key prefix, mask, degrees) of transform t's masked linears, as nflows' MADE builds them with
        random_mask=False (the only form `build_maf_rqs` produces, flow.py:291-308)."""
        h = self.hyper
        first = 1 if self.z_score_theta else 0
        pre = f"_transform._transforms.{first + 2 * t}.autoregressive_net."
        hd = h.hidden_degrees()
        od = torch.repeat_interleave(torch.arange(1, h.D + 1), 3 * h.num_bins - 1)
        yield pre + "initial_layer.", h.mask(0), hd.clone()
        for b in range(h.num_blocks):
            yield pre + f"blocks.{b}.linear.", h.mask(2), hd.clone()
        yield pre + "final_layer.", h.mask(3), od

    @torch.no_grad()
    def load_nflows_state_dict(self, sd: Dict[str, Tensor], prefix: str = "net.") -> None:
        h = self.hyper
        first = 1 if self.z_score_theta else 0
        # a checkpoint whose masks differ from the degree rule the kernels evaluate (random_mask=True, a different
        # nflows version) must not load silently
        for t in range(h.num_transforms):
            for key, mask, _ in self._mask_buffers(t):
                got = sd.get(prefix + key + "mask")
                if got is not None and not torch.equal(got.to(torch.float32).cpu(), mask):
                    raise ValueError(f"{key}mask differs from the MADE degree masks the maf_rqs kernels apply "
                                     "(random_mask=True checkpoints are not supported)")
        for key, off, n, shape in self._slices():
            src = sd[prefix + key]
            if tuple(src.shape) != tuple(shape):
                raise ValueError(f"{key}: expected {shape}, got {tuple(src.shape)}")
            self.flat_params[off : off + n].copy_(src.reshape(-1).to(self.flat_params))
        for t in range(h.num_transforms):
            self.perms[t].copy_(sd[prefix + f"_transform._transforms.{first + 2 * t + 1}._permutation"].to(torch.int32))
        if self.z_score_theta:
            self.zstats[: h.D].copy_(sd[prefix + "_transform._transforms.0._shift"].reshape(-1))
            self.zstats[h.D : 2 * h.D].copy_(sd[prefix + "_transform._transforms.0._scale"].reshape(-1))
        if self.z_score_x:
            self.zstats[2 * h.D : 2 * h.D + h.C].copy_(sd[prefix + "_embedding_net.0._mean"].reshape(-1).expand(h.C))
            self.zstats[2 * h.D + h.C :].copy_(sd[prefix + "_embedding_net.0._std"].reshape(-1).expand(h.C))
        self.__dict__.pop("_packed_cache", None)

    def kernel_masks(self) -> Optional[Tensor]:
        """Mask buffer handed to the kernels (None: the MADE degree formulas are evaluated on the device)."""
        return None

    # -- fused training pass -----------------------------------------------------------------------
    def train_workspace_floats(self, n: int) -> int:
        need = _lib.load().sbi_amd_maf_train_workspace_floats(self.hyper.c_config(), n)
        if need < 0:
            _lib.check(int(need), "maf_train_workspace_floats")
        return int(need)

    def train_pass(self, theta: Tensor, x: Tensor, row_weight: Optional[Tensor], uniform_weight: float,
                   grad_out: Tensor, workspace: Optional[Tensor] = None, want_grad_theta: bool = False,
                   grad_x_out: Optional[Tensor] = None):
        if grad_x_out is not None:
            raise NotImplementedError("maf_rqs kernels do not return d loss / d embedded x (no trainable embedding)")
        return maf_loss_fwd_bwd(self, theta, x, row_weight, uniform_weight, grad_out, want_grad_theta, workspace)


# --------------------------------------------------------------------- kernel calls
def maf_packed_weights(net: MAFNet) -> Tensor:
    fp = net.flat_params
    key = (fp.data_ptr(), fp._version, str(fp.device), net.perms._version)
    cache = net.__dict__.get("_packed_cache")
    if cache is not None and cache[0] == key:
        return cache[1]
    dev = _lib.require_device(fp)
    lib = _lib.load()
    cfg = net.hyper.c_config()
    n = lib.sbi_amd_maf_packed_floats(cfg)
    if n < 0:
        _lib.check(int(n), "maf_packed_floats")
    packed = cache[1] if (cache is not None and cache[1].device == dev and cache[1].numel() == n) else \
        torch.zeros(int(n), dtype=torch.float32, device=dev)
    perms = net.perms.contiguous()
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_maf_pack(cfg, _lib.ptr(fp), perms.data_ptr(), _lib.ptr(net.kernel_masks()), _lib.ptr(packed),
                                  _lib.current_stream(dev))
    _lib.check(rc, "maf_pack")
    net.__dict__["_packed_cache"] = (key, packed)
    return packed


def maf_log_prob_call(net: MAFNet, theta: Tensor, x: Tensor, want_noise: bool):
    dev = _lib.require_device(theta, x, net.flat_params, net.zstats)
    lib = _lib.load()
    n = theta.shape[0]
    logp = torch.empty(n, dtype=torch.float32, device=dev)
    noise = torch.empty_like(theta) if want_noise else None
    if n == 0:
        return logp, noise
    packed = maf_packed_weights(net)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_maf_log_prob(net.hyper.c_config(), _lib.ptr(packed), _lib.ptr(net.zstats), _lib.ptr(theta),
                                      _lib.ptr(x), n, x.shape[0], _lib.ptr(logp), _lib.ptr(noise),
                                      _lib.current_stream(dev))
    _lib.check(rc, "maf_log_prob")
    return logp, noise


def maf_sample_call(net: MAFNet, noise: Tensor, x: Tensor, want_ld: bool):
    dev = _lib.require_device(noise, x, net.flat_params, net.zstats)
    lib = _lib.load()
    n = noise.shape[0]
    theta = torch.empty_like(noise)
    ld = torch.empty(n, dtype=torch.float32, device=dev) if want_ld else None
    if n == 0:
        return theta, ld
    packed = maf_packed_weights(net)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_maf_sample(net.hyper.c_config(), _lib.ptr(packed), _lib.ptr(net.zstats), _lib.ptr(noise),
                                    _lib.ptr(x), n, x.shape[0], _lib.ptr(theta), _lib.ptr(ld),
                                    _lib.current_stream(dev))
    _lib.check(rc, "maf_sample")
    return theta, ld


def maf_loss_fwd_bwd(net: MAFNet, theta: Tensor, x: Tensor, row_weight: Optional[Tensor], uniform_weight: float,
                     grad_out: Tensor, want_grad_theta: bool = False, workspace: Optional[Tensor] = None):
    """Fused training pass: (per-row loss, grad_theta | None); fills grad_out (P,)."""
    dev = _lib.require_device(theta, x, net.flat_params, net.zstats, grad_out, row_weight)
    lib = _lib.load()
    n = theta.shape[0]
    need = net.train_workspace_floats(n)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
    loss = torch.empty(n, dtype=torch.float32, device=dev)
    gtheta = torch.empty_like(theta) if want_grad_theta else None
    packed = maf_packed_weights(net)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_maf_loss_fwd_bwd(net.hyper.c_config(), _lib.ptr(packed), _lib.ptr(net.zstats),
                                          _lib.ptr(net.kernel_masks()), _lib.ptr(theta), _lib.ptr(x), n, x.shape[0],
                                          _lib.ptr(row_weight),
                                          float(uniform_weight), _lib.ptr(loss), _lib.ptr(grad_out), _lib.ptr(gtheta),
                                          _lib.ptr(workspace), _lib.current_stream(dev))
    _lib.check(rc, "maf_loss_fwd_bwd")
    return loss, gtheta


class _MAFLogProbFn(torch.autograd.Function):
    """Autograd bridge: forward = the log_prob kernel; backward = the fused training pass with row weights
    -dL/dlogp (it re-runs the forward with the per-transform stash: the maf_rqs training pass is one call)."""

    @staticmethod
    def forward(ctx, theta: Tensor, x: Tensor, flat_params: Tensor, net: MAFNet):
        ctx.net = net
        ctx.version = net.flat_params._version
        ctx.save_for_backward(theta, x)
        logp, _ = maf_log_prob_call(net, theta, x, want_noise=False)
        return logp

    @staticmethod
    def backward(ctx, grad_logp: Tensor):
        theta, x = ctx.saved_tensors
        net: MAFNet = ctx.net
        if net.flat_params._version != ctx.version:
            raise RuntimeError("maf_rqs parameters were modified in place between log_prob() and backward().")
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("maf_rqs kernels do not return the gradient wrt the condition")
        gparams = torch.empty_like(net.flat_params)
        w = (-grad_logp).contiguous().to(torch.float32)
        _, gtheta = maf_loss_fwd_bwd(net, theta, x, w, 0.0, gparams, want_grad_theta=ctx.needs_input_grad[0])
        return gtheta, None, (gparams if ctx.needs_input_grad[2] else None), None


class MAFRQSFlow(NSFFlow):
    r"""Masked autoregressive RQ-spline flow :math:`p(\theta|x)` evaluated by the gfx950 kernels."""

    def __init__(self, net: MAFNet, input_shape: torch.Size, condition_shape: torch.Size,
                 embedding_net: Optional[nn.Module] = None) -> None:
        super().__init__(net, input_shape=input_shape, condition_shape=condition_shape, embedding_net=embedding_net)
        if embedding_net is not None and any(p.requires_grad for p in embedding_net.parameters()):
            raise NotImplementedError("sbi_amd maf_rqs: trainable embedding nets are not supported (frozen / "
                                      "parameter-free ones are applied in front of the kernels)")

    def _raw_log_prob(self, net, theta: Tensor, x: Tensor, want_noise: bool):
        return maf_log_prob_call(net, theta, x, want_noise)

    def _raw_sample(self, net, noise: Tensor, x: Tensor, want_ld: bool):
        return maf_sample_call(net, noise, x, want_ld)

    def _raw_autograd(self, net, theta: Tensor, x: Tensor, flat: Tensor) -> Tensor:
        return _MAFLogProbFn.apply(theta, x, flat, net)
