"""NSF density estimator on the MI355X HIP kernels.

``NSFFlow`` is the drop-in for sbi's ``NFlowsFlow(build_nsf(...))``
(sbi/neural_nets/estimators/nflows_flow.py:14-151): same constructor role,
``log_prob`` / ``loss`` / ``sample`` / ``sample_and_log_prob`` /
``inverse_transform`` with the same shapes.  All arithmetic runs in
``libsbi_amd_nsf.so`` through the C ABI of include/sbi_amd_nsf.h; there is no
PyTorch or CPU fallback.

Parameters live in ONE flat fp32 ``nn.Parameter`` (``net.flat_params``) laid
out in nflows' natural order, so Adam state, gradient clipping and the
data-parallel all-reduce each touch a single contiguous buffer.  Weight
exchange with a real nflows ``Flow`` goes through ``nflows_state_dict`` /
``load_nflows_state_dict`` (key names of SURVEY.md Appendix C).
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
from sbi_amd.neural_nets.estimators.base import ConditionalDensityEstimator


@dataclass(frozen=True)
class NSFHyper:
    """Hyper-parameters ``build_nsf`` bakes into the flow (flow.py:333-352)."""

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
    lu_eps: float = 1e-3
    hidden_layers_spline_context: int = 1   # theta-dim 1 only: applications of ContextSplineMap's ONE hidden Linear

    def c_config(self) -> _lib.NSFConfigC:
        return _lib.NSFConfigC(
            self.D, self.C, self.hidden_features, self.num_bins, self.num_transforms, self.num_blocks,
            self.tail_bound, self.min_bin_width, self.min_bin_height, self.min_derivative, self.lu_eps,
            (int(self.hidden_layers_spline_context) or -1) if self.D == 1 else 0,     # C side: 0 = default, -1 = none
        )

    # -- layout of the flat buffer (must agree with csrc/nsf_plan.cpp) -----------
    @property
    def ctx_mlp(self) -> bool:
        """theta-dim 1: sbi swaps the ResidualNet for ``ContextSplineMap`` (flow.py:401-408), uses the
        dummy mask [1] in every transform and drops the LULinear layers (flow.py:426, 436)."""
        return self.D == 1

    def d_tr(self, t: int) -> int:
        if self.ctx_mlp:
            return 1
        return (self.D + 1) // 2 if t % 2 == 0 else self.D // 2

    def d_id(self, t: int) -> int:
        return self.D - self.d_tr(t)

    def layer_entries(self, t: int) -> List[Tuple[str, Tuple[int, ...]]]:
        """(nflows sub-key, shape) in flat order for transform t."""
        H, C, P = self.hidden_features, self.C, 3 * self.num_bins - 1
        if self.ctx_mlp:
            # nn.Sequential(Linear, ReLU, [Linear, ReLU] * n, Linear): the SAME hidden Linear object sits at indices
            # 2, 4, ..., 2 n (flow.py:1456-1462); it is stored once (under index 2), the output layer is index 2 + 2 n
            pre = "transform_net.spline_predictor."
            fin = 2 + 2 * self.hidden_layers_spline_context
            hidden = [(pre + "2.weight", (H, H)), (pre + "2.bias", (H,))] if self.hidden_layers_spline_context > 0 else []
            return [(pre + "0.weight", (H, C)), (pre + "0.bias", (H,))] + hidden + \
                   [(pre + f"{fin}.weight", (P, H)), (pre + f"{fin}.bias", (P,))]
        out = [("transform_net.initial_layer.weight", (H, self.d_id(t) + C)),
               ("transform_net.initial_layer.bias", (H,))]
        for b in range(self.num_blocks):
            pre = f"transform_net.blocks.{b}."
            out += [(pre + "context_layer.weight", (H, C)), (pre + "context_layer.bias", (H,)),
                    (pre + "linear_layers.0.weight", (H, H)), (pre + "linear_layers.0.bias", (H,)),
                    (pre + "linear_layers.1.weight", (H, H)), (pre + "linear_layers.1.bias", (H,))]
        out += [("transform_net.final_layer.weight", (self.d_tr(t) * P, H)),
                ("transform_net.final_layer.bias", (self.d_tr(t) * P,))]
        return out

    def lu_entries(self) -> List[Tuple[str, Tuple[int, ...]]]:
        if self.ctx_mlp:
            return []
        n_tri = self.D * (self.D - 1) // 2
        return [("lower_entries", (n_tri,)), ("upper_entries", (n_tri,)),
                ("unconstrained_upper_diag", (self.D,)), ("bias", (self.D,))]

    def param_count(self) -> int:
        n = 0
        for t in range(self.num_transforms):
            n += sum(int(np.prod(s)) for _, s in self.layer_entries(t))
            n += sum(int(np.prod(s)) for _, s in self.lu_entries())
        return n


class NSFNet(nn.Module):
    """Parameter/buffer holder playing the role of nflows' ``Flow`` object.

    Buffers: ``zstats`` = [theta shift (D), theta scale (D), x mean (C), x std (C)]
    (PointwiseAffineTransform / Standardize buffers, sbiutils.py:226-247, 418-428),
    ``_log_z`` (flow.py:1481-1488).
    """

    def __init__(self, hyper: NSFHyper, zstats: Tensor, z_score_theta: bool, z_score_x: bool,
                 dtype: torch.dtype = torch.float32):
        super().__init__()
        self.hyper = hyper
        self.z_score_theta = z_score_theta
        self.z_score_x = z_score_x
        self.flat_params = nn.Parameter(torch.zeros(hyper.param_count(), dtype=torch.float32))
        self.register_buffer("zstats", zstats.to(torch.float32).contiguous())
        self.register_buffer(
            "_log_z", torch.tensor(0.5 * hyper.D * math.log(2 * math.pi), dtype=torch.float64).to(dtype),
            persistent=False,
        )
        self.reset_parameters()
        # Checkpoints are interchangeable with the reference's: `state_dict()` speaks nflows' key names (what
        # `deepcopy(neural_net.state_dict())` / `load_state_dict` of sbi's trainer and users' save / load code see,
        # trainers/base.py:1275-1281, tests/save_and_load_test.py:23-43) and `load_state_dict()` accepts them -- as
        # well as the native two-tensor form (`flat_params`, `zstats`) of earlier versions of this package.
        self._register_state_dict_hook(NSFNet._emit_nflows_keys)
        self._register_load_state_dict_pre_hook(self._accept_nflows_keys, with_module=False)

    _NATIVE_KEYS = ("flat_params", "zstats")

    @staticmethod
    def _emit_nflows_keys(module, state_dict, prefix, local_metadata):
        """state_dict hook: replace the native entries by the reference's keys."""
        if getattr(module, "_native_state_dict", False):
            return state_dict
        for k in NSFNet._NATIVE_KEYS:
            state_dict.pop(prefix + k, None)
        for k, v in module.nflows_state_dict(prefix=prefix, with_masks=True).items():
            state_dict[k] = v
        return state_dict

    def _accept_nflows_keys(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """load_state_dict pre-hook: nflows-keyed entries under `prefix` become the native pair."""
        if prefix + "flat_params" in state_dict:
            return      # native form
        mine = {k: v for k, v in state_dict.items() if k.startswith(prefix + "_transform.") or k.startswith(prefix + "_embedding_net.0._")
                or k.startswith(prefix + "_distribution.")}
        if not mine:
            return      # nothing of ours: let the regular missing-key report speak
        flat = self.flat_params.detach().clone()
        zst = self.zstats.detach().clone()
        try:
            self._fill_from_nflows(mine, prefix, flat, zst)
        except (KeyError, ValueError) as e:
            error_msgs.append(f"NSFNet: cannot read the nflows-keyed checkpoint: {e!r}")
            return
        for k in mine:
            del state_dict[k]
        state_dict[prefix + "flat_params"] = flat
        state_dict[prefix + "zstats"] = zst

    def native_state_dict(self) -> "OrderedDict[str, Tensor]":
        """The two-tensor form (`flat_params`, `zstats`): what the kernels read, no per-layer views."""
        self._native_state_dict = True
        try:
            return self.state_dict()
        finally:
            self._native_state_dict = False

    # nflows-equivalent initialisation, drawing from torch's global generator in
    # the same order nflows constructs its modules.
    @torch.no_grad()
    def reset_parameters(self) -> None:
        h = self.hyper
        chunks: List[Tensor] = []
        for t in range(h.num_transforms):
            mods: Dict[str, nn.Linear] = {}
            if h.ctx_mlp:
                pre = "transform_net.spline_predictor."
                mods[pre + "0"] = nn.Linear(h.C, h.hidden_features)
                # (`[nn.Linear(H, H), nn.ReLU()] * 0` still CONSTRUCTS the Linear -- and draws its initial weights --
                #  before the empty repetition discards it, flow.py:1457-1460: same generator order here)
                mods[pre + "2"] = nn.Linear(h.hidden_features, h.hidden_features)
                mods[pre + str(2 + 2 * h.hidden_layers_spline_context)] = nn.Linear(h.hidden_features, 3 * h.num_bins - 1)
                for key, _shape in h.layer_entries(t):
                    mod, attr = key.rsplit(".", 1)
                    chunks.append(getattr(mods[mod], attr).detach().reshape(-1))
                continue
            mods["transform_net.initial_layer"] = nn.Linear(h.d_id(t) + h.C, h.hidden_features)
            for b in range(h.num_blocks):
                pre = f"transform_net.blocks.{b}."
                mods[pre + "context_layer"] = nn.Linear(h.C, h.hidden_features)
                mods[pre + "linear_layers.0"] = nn.Linear(h.hidden_features, h.hidden_features)
                last = nn.Linear(h.hidden_features, h.hidden_features)
                nn.init.uniform_(last.weight, -1e-3, 1e-3)   # ResidualBlock zero_initialization
                nn.init.uniform_(last.bias, -1e-3, 1e-3)
                mods[pre + "linear_layers.1"] = last
            mods["transform_net.final_layer"] = nn.Linear(h.hidden_features, h.d_tr(t) * (3 * h.num_bins - 1))
            for key, _shape in h.layer_entries(t):
                mod, attr = key.rsplit(".", 1)
                chunks.append(getattr(mods[mod], attr).detach().reshape(-1))
            n_tri = h.D * (h.D - 1) // 2
            chunks.append(torch.zeros(2 * n_tri))                                   # L, U strict entries
            chunks.append(torch.full((h.D,), float(np.log(np.exp(1 - h.lu_eps) - 1))))  # identity init
            chunks.append(torch.zeros(h.D))                                         # bias
        flat = torch.cat(chunks)
        assert flat.numel() == self.flat_params.numel()
        self.flat_params.copy_(flat)

    # -- fused training pass (what FusedTrainStep drives) --------------------------------
    supports_atomic = True     # train_forward / train_backward on one stash (multi-round NPE-C)

    def train_workspace_floats(self, n: int) -> int:
        need = _lib.load().sbi_amd_nsf_train_workspace_floats(self.hyper.c_config(), n)
        if need < 0:
            _lib.check(int(need), "nsf_train_workspace_floats")
        return int(need)

    def train_pass(self, theta: Tensor, x: Tensor, row_weight: Optional[Tensor], uniform_weight: float,
                   grad_out: Tensor, workspace: Optional[Tensor] = None, want_grad_theta: bool = False,
                   grad_x_out: Optional[Tensor] = None):
        return loss_fwd_bwd(self, theta, x, row_weight, uniform_weight, grad_out, want_grad_theta, workspace,
                            grad_x_out)

    # -- nflows state_dict exchange -------------------------------------------------
    def _slices(self):
        h = self.hyper
        first = 1 if self.z_score_theta else 0
        off = 0
        stride = 1 if h.ctx_mlp else 2   # no LULinear between couplings for theta-dim 1
        for t in range(h.num_transforms):
            pre = f"_transform._transforms.{first + stride * t}."
            for key, shape in h.layer_entries(t):
                n = int(np.prod(shape))
                yield pre + key, off, n, shape
                off += n
            pre = f"_transform._transforms.{first + stride * t + 1}."
            for key, shape in h.lu_entries():
                n = int(np.prod(shape))
                yield pre + key, off, n, shape
                off += n

    def nflows_state_dict(self, prefix: str = "net.", with_masks: bool = False) -> "OrderedDict[str, Tensor]":
        h = self.hyper
        sd: "OrderedDict[str, Tensor]" = OrderedDict()
        flat = self.flat_params.detach()
        if self.z_score_theta:
            sd[prefix + "_transform._transforms.0._shift"] = self.zstats[: h.D].clone()
            sd[prefix + "_transform._transforms.0._scale"] = self.zstats[h.D : 2 * h.D].clone()
        seen_masks = set()
        for key, off, n, shape in self._slices():
            if with_masks and ".transform_net." in key:
                # the coupling transforms' index buffers (nflows CouplingTransform registers them): transform t masks
                # with create_alternating_binary_mask(D, even = t % 2 == 0) (flow.py:1395-1416); theta-dim 1: mask [1]
                tpre = key.split(".transform_net.")[0]
                if tpre not in seen_masks:
                    seen_masks.add(tpre)
                    t = len(seen_masks) - 1
                    dev = flat.device
                    if h.ctx_mlp:
                        ident, trans = torch.zeros(0, dtype=torch.long, device=dev), torch.zeros(1, dtype=torch.long, device=dev)
                    else:
                        trans = torch.arange(t % 2, h.D, 2, dtype=torch.long, device=dev)
                        ident = torch.arange(1 - t % 2, h.D, 2, dtype=torch.long, device=dev)
                    sd[prefix + tpre + ".identity_features"] = ident
                    sd[prefix + tpre + ".transform_features"] = trans
            sd[prefix + key] = flat[off : off + n].reshape(shape).clone()
            if h.ctx_mlp and ".spline_predictor.2." in key:
                # the reference's state_dict lists the shared hidden Linear under every index it occupies
                for i in range(2, h.hidden_layers_spline_context + 1):
                    sd[prefix + key.replace(".spline_predictor.2.", f".spline_predictor.{2 * i}.")] = sd[prefix + key].clone()
        if self.z_score_x:
            sd[prefix + "_embedding_net.0._mean"] = self.zstats[2 * h.D : 2 * h.D + h.C].clone()
            sd[prefix + "_embedding_net.0._std"] = self.zstats[2 * h.D + h.C :].clone()
        return sd

    @torch.no_grad()
    def load_nflows_state_dict(self, sd: Dict[str, Tensor], prefix: str = "net.") -> None:
        self._fill_from_nflows(sd, prefix, self.flat_params, self.zstats)

    @torch.no_grad()
    def _fill_from_nflows(self, sd: Dict[str, Tensor], prefix: str, flat_params: Tensor, zstats: Tensor) -> None:
        h = self.hyper
        for key, off, n, shape in self._slices():
            src = sd[prefix + key]
            if tuple(src.shape) != tuple(shape):
                raise ValueError(f"{key}: expected {shape}, got {tuple(src.shape)}")
            if h.ctx_mlp and ".spline_predictor.2." in key:      # one module under several indices: they must agree
                for i in range(2, h.hidden_layers_spline_context + 1):
                    dup = sd.get(prefix + key.replace(".spline_predictor.2.", f".spline_predictor.{2 * i}."))
                    if dup is not None and not torch.equal(dup, src):
                        raise ValueError(f"{key}: the checkpoint holds different weights for the repeated hidden layer "
                                         "(ContextSplineMap shares ONE Linear across hidden_layers_spline_context)")
            flat_params[off : off + n].copy_(src.reshape(-1).to(flat_params))
        if self.z_score_theta:
            zstats[: h.D].copy_(sd[prefix + "_transform._transforms.0._shift"].reshape(-1))
            zstats[h.D : 2 * h.D].copy_(sd[prefix + "_transform._transforms.0._scale"].reshape(-1))
        if self.z_score_x:
            mean = sd[prefix + "_embedding_net.0._mean"].reshape(-1)
            std = sd[prefix + "_embedding_net.0._std"].reshape(-1)
            zstats[2 * h.D : 2 * h.D + h.C].copy_(mean.expand(h.C))
            zstats[2 * h.D + h.C :].copy_(std.expand(h.C))


# --------------------------------------------------------------------- kernel calls
def packed_weights(net: NSFNet, rows: Optional[int] = None, training: bool = False, sampling: bool = False) -> Tensor:
    """Kernel-side weight images of ``net.flat_params`` (re-packed lazily when the parameter tensor was modified or
    moved; tracked through its version counter).  The buffer holds two images (include/sbi_amd_nsf.h): the throughput
    kernels' and the cooperative small-batch kernels'; a call that knows its row count (`rows`) only re-packs the one
    its kernels read -- a training loop at a fixed batch size pays for one image per step, not two."""
    fp = net.flat_params
    key = (fp.data_ptr(), fp._version, str(fp.device))
    cache = net.__dict__.get("_packed_cache")
    lib = _lib.load()
    cfg = net.hyper.c_config()
    # bits: 1 throughput image, 2 cooperative image, 8 / 4 their explicit LU inverses (read by the sampling direction
    # only: a serial fp64 substitution per transform, not paid per training step)
    want = 15 if rows is None else (2 if lib.sbi_amd_nsf_image_kind(cfg, int(rows), int(training)) == 1 else 1)
    if sampling:
        want |= 4 if want == 2 else 8
    have = 0
    if cache is not None and cache[0] == key:
        have = net.__dict__.get("_packed_images", 0)
        if have & want == want:
            return cache[1]
    dev = _lib.require_device(fp)
    n = lib.sbi_amd_nsf_packed_floats(cfg)
    if n < 0:
        _lib.check(int(n), "nsf_packed_floats")
    if cache is not None and cache[1].device == dev and cache[1].numel() == n:
        packed = cache[1]
    else:
        packed, have = torch.zeros(int(n), dtype=torch.float32, device=dev), 0   # (gaps of the images are never written)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_nsf_pack_images(cfg, _lib.ptr(fp), _lib.ptr(packed), want & ~have, _lib.current_stream(dev))
    _lib.check(rc, "nsf_pack")
    net.__dict__["_packed_cache"] = (key, packed)
    net.__dict__["_packed_images"] = have | want
    return packed


def _log_prob_call(net: NSFNet, theta: Tensor, x: Tensor, want_noise: bool) -> Tuple[Tensor, Optional[Tensor]]:
    """theta (N,D), x (x_rows,C) -> logp (N,), noise (N,D)|None."""
    dev = _lib.require_device(theta, x, net.flat_params, net.zstats)
    lib = _lib.load()
    n = theta.shape[0]
    logp = torch.empty(n, dtype=torch.float32, device=dev)
    noise = torch.empty_like(theta) if want_noise else None
    if n == 0:
        return logp, noise
    packed = packed_weights(net, rows=n)
    cfg = net.hyper.c_config()
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_nsf_log_prob(
            cfg, _lib.ptr(packed), _lib.ptr(net.zstats), _lib.ptr(theta), _lib.ptr(x), n, x.shape[0],
            _lib.ptr(logp), _lib.ptr(noise), _lib.current_stream(dev),
        )
    _lib.check(rc, "nsf_log_prob")
    return logp, noise


def _sample_call(net: NSFNet, noise: Tensor, x: Tensor, want_ld: bool) -> Tuple[Tensor, Optional[Tensor]]:
    dev = _lib.require_device(noise, x, net.flat_params, net.zstats)
    lib = _lib.load()
    n = noise.shape[0]
    theta = torch.empty_like(noise)
    ld = torch.empty(n, dtype=torch.float32, device=dev) if want_ld else None
    if n == 0:
        return theta, ld
    cfg = net.hyper.c_config()
    # (small calls and every call of a net with hidden > 64 run on the cooperative kernels, the rest on the throughput
    #  kernel: sbi_amd_nsf_image_kind answers for the sampling direction as it does for log_prob; + the LU inverses)
    packed = packed_weights(net, rows=n, sampling=True)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_nsf_sample(
            cfg, _lib.ptr(packed), _lib.ptr(net.zstats), _lib.ptr(noise), _lib.ptr(x), n, x.shape[0],
            _lib.ptr(theta), _lib.ptr(ld), _lib.current_stream(dev),
        )
    _lib.check(rc, "nsf_sample")
    return theta, ld


def loss_fwd_bwd(net: NSFNet, theta: Tensor, x: Tensor, row_weight: Optional[Tensor], uniform_weight: float,
                 grad_out: Tensor, want_grad_theta: bool = False, workspace: Optional[Tensor] = None,
                 grad_x_out: Optional[Tensor] = None):
    """Fused training pass: returns (per-row loss, grad_theta|None); fills grad_out (P,) and, if given, grad_x_out
    (n, C: one context row per theta row)."""
    dev = _lib.require_device(theta, x, net.flat_params, net.zstats, grad_out, row_weight)
    lib = _lib.load()
    n = theta.shape[0]
    cfg = net.hyper.c_config()
    need = lib.sbi_amd_nsf_train_workspace_floats(cfg, n)
    if need < 0:
        _lib.check(int(need), "nsf_train_workspace_floats")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(int(need), 1), dtype=torch.float32, device=dev)
    loss = torch.empty(n, dtype=torch.float32, device=dev)
    gtheta = torch.empty_like(theta) if want_grad_theta else None
    packed = packed_weights(net, rows=n, training=True)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_nsf_loss_fwd_bwd(
            cfg, _lib.ptr(net.flat_params), _lib.ptr(packed), _lib.ptr(net.zstats), _lib.ptr(theta), _lib.ptr(x), n, x.shape[0],
            _lib.ptr(row_weight), float(uniform_weight), _lib.ptr(loss), _lib.ptr(grad_out), _lib.ptr(gtheta),
            _lib.ptr(grad_x_out), _lib.ptr(workspace), _lib.current_stream(dev),
        )
    _lib.check(rc, "nsf_loss_fwd_bwd")
    return loss, gtheta


def train_workspace(net: NSFNet, n: int, device, workspace: Optional[Tensor] = None) -> Tensor:
    """A workspace tensor large enough for an n-row training pass (reuses `workspace` when it is)."""
    lib = _lib.load()
    need = lib.sbi_amd_nsf_train_workspace_floats(net.hyper.c_config(), n)
    if need < 0:
        _lib.check(int(need), "nsf_train_workspace_floats")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(int(need), 1), dtype=torch.float32, device=device)
    return workspace


def train_forward(net: NSFNet, theta: Tensor, x: Tensor, workspace: Tensor) -> Tensor:
    """First half of the training pass: log p (n,) with the state / activation stash left in `workspace`.
    `x` may have fewer rows than theta: row r is conditioned on x[r % x.shape[0]]."""
    dev = _lib.require_device(theta, x, net.flat_params, net.zstats, workspace)
    lib = _lib.load()
    n = theta.shape[0]
    logp = torch.empty(n, dtype=torch.float32, device=dev)
    packed = packed_weights(net, rows=n, training=True)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_nsf_train_forward(
            net.hyper.c_config(), _lib.ptr(packed), _lib.ptr(net.zstats), _lib.ptr(theta), _lib.ptr(x), n, x.shape[0],
            _lib.ptr(logp), _lib.ptr(workspace), _lib.current_stream(dev),
        )
    _lib.check(rc, "nsf_train_forward")
    return logp


def train_backward(net: NSFNet, x: Tensor, n: int, row_weight: Tensor, grad_out: Tensor, workspace: Tensor,
                   want_grad_theta: bool = False, grad_x_out: Optional[Tensor] = None) -> Optional[Tensor]:
    """Second half: grad_out (P,) = d( sum_n w_n * (-log p_n) ) / d params from the stash `train_forward` left."""
    dev = _lib.require_device(x, net.flat_params, net.zstats, grad_out, row_weight, workspace)
    lib = _lib.load()
    gtheta = torch.empty(n, net.hyper.D, dtype=torch.float32, device=dev) if want_grad_theta else None
    packed = packed_weights(net, rows=n, training=True)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_nsf_train_backward(
            net.hyper.c_config(), _lib.ptr(net.flat_params), _lib.ptr(packed), _lib.ptr(net.zstats), _lib.ptr(x), n,
            x.shape[0], _lib.ptr(row_weight), 0.0, _lib.ptr(grad_out), _lib.ptr(gtheta), _lib.ptr(grad_x_out),
            _lib.ptr(workspace), _lib.current_stream(dev),
        )
    _lib.check(rc, "nsf_train_backward")
    return gtheta


class _NSFLogProbFn(torch.autograd.Function):
    """Autograd bridge: forward = the training pass's forward half (log p + state / activation stash kept in a
    workspace tensor on the graph), backward = the backward half with row weights -dL/dlogp on that stash --
    so `loss.backward()` in sbi's own loop costs one forward and one backward, not two forwards."""

    @staticmethod
    def forward(ctx, theta: Tensor, x: Tensor, flat_params: Tensor, net: NSFNet):
        ctx.net = net
        ctx.n = theta.shape[0]
        ctx.packed_version = net.flat_params._version
        lib = _lib.load()
        if ctx.n == 0 or lib.sbi_amd_nsf_train_workspace_floats(net.hyper.c_config(), max(ctx.n, 1)) < 0:
            # configurations the training kernels do not cover (e.g. hidden_features = 64) still evaluate;
            # only an actual backward() through them reports the restriction
            logp, _ = _log_prob_call(net, theta, x, want_noise=False)
            ctx.stashed = False
            ctx.save_for_backward(x, theta)
            return logp
        ws = train_workspace(net, ctx.n, theta.device)
        logp = train_forward(net, theta, x, ws)
        ctx.stashed = True
        ctx.save_for_backward(x, ws)
        return logp

    @staticmethod
    def backward(ctx, grad_logp: Tensor):
        x, ws = ctx.saved_tensors
        net: NSFNet = ctx.net
        if net.flat_params._version != ctx.packed_version:
            raise RuntimeError("NSF parameters were modified in place between log_prob() and backward().")
        gparams = torch.empty_like(net.flat_params)
        w = (-grad_logp).contiguous().to(torch.float32)
        gx = None
        if ctx.needs_input_grad[1]:
            if x.shape[0] != ctx.n:
                raise RuntimeError("gradient wrt the condition needs one condition row per theta row")
            gx = torch.empty_like(x)
        if not ctx.stashed:   # ws holds theta here: the one-call form raises the configuration error
            _, gtheta = loss_fwd_bwd(net, ws, x, w, 0.0, gparams, want_grad_theta=ctx.needs_input_grad[0],
                                     grad_x_out=gx)
            return gtheta, gx, (gparams if ctx.needs_input_grad[2] else None), None
        # kernel gradients are already weighted by w_n = -dL/dlogp_n
        gtheta = train_backward(net, x, ctx.n, w, gparams, ws, want_grad_theta=ctx.needs_input_grad[0],
                                grad_x_out=gx)
        return gtheta, gx, (gparams if ctx.needs_input_grad[2] else None), None


class NSFFlow(ConditionalDensityEstimator):
    r"""Neural Spline Flow :math:`p(\theta|x)` evaluated by the gfx950 kernels."""

    def __init__(self, net: NSFNet, input_shape: torch.Size, condition_shape: torch.Size,
                 embedding_net: Optional[nn.Module] = None) -> None:
        super().__init__(net, input_shape=input_shape, condition_shape=condition_shape)
        if len(torch.Size(input_shape)) != 1 or (len(torch.Size(condition_shape)) != 1 and embedding_net is None):
            raise NotImplementedError(
                "sbi_amd NSF kernels take 1-D theta events and 1-D (embedded) x features; pass an embedding_net "
                "for structured x."
            )
        self.net: NSFNet
        # `standardizing_net -> embedding_net` in front of the flow (flow.py:1395-1416): ordinary PyTorch; the
        # kernels consume its (B, C) output and return d loss / d output for autograd
        self._embedding_net = embedding_net

    @property
    def embedding_net(self) -> nn.Module:
        return nn.Identity() if self._embedding_net is None else self._embedding_net

    @property
    def _cdim(self) -> int:
        """feature width the kernels see"""
        return self.net.hyper.C

    def _embed(self, condition: Tensor) -> Tensor:
        """(..., *condition_shape) -> (..., C)"""
        if self._embedding_net is None:
            return condition
        nd = len(self.condition_shape)
        lead = condition.shape[: condition.dim() - nd]
        e = self._embedding_net(condition.reshape(-1, *self.condition_shape).float())
        return e.reshape(*lead, self._cdim)

    # -- kernel hooks (the maf_rqs / zuko estimators bind their own C entry points in the `_raw_*` three) -----
    def _raw_log_prob(self, net, theta: Tensor, x: Tensor, want_noise: bool):
        return _log_prob_call(net, theta, x, want_noise)

    def _raw_sample(self, net, noise: Tensor, x: Tensor, want_ld: bool):
        return _sample_call(net, noise, x, want_ld)

    def _raw_autograd(self, net, theta: Tensor, x: Tensor, flat: Tensor) -> Tensor:
        return _NSFLogProbFn.apply(theta, x, flat, net)

    # -- host residency: a CPU-resident estimator answers through the current ROCm device ------------------------
    def _kernel_net(self):
        """The net whose buffers the kernels read: `self.net` when it lives on a ROCm device; for a CPU-resident
        estimator (sbi builds on the CPU and probes `log_prob` there before `.to(device)`:
        npe_base.py:694-703, user_input_checks.py:767-795) a device mirror of its 392 KB of parameters and z-score
        buffers, refreshed whenever the host tensors changed.  The arithmetic is the HIP kernels' either way --
        there is no CPU implementation to fall back to, so without a ROCm device this raises."""
        fp = self.net.flat_params
        if fp.is_cuda:
            self.__dict__.pop("_mirror", None)
            return self.net
        if not torch.cuda.is_available():
            raise RuntimeError(
                "sbi_amd: the NSF hot path runs only on a ROCm device (MI355X) and none is visible; a CPU-resident "
                "estimator is evaluated by staging through the device. There is deliberately no CPU fallback.")
        dev = torch.device("cuda", torch.cuda.current_device())
        key = (fp.data_ptr(), fp._version, self.net.zstats._version, dev.index)
        held = self.__dict__.get("_mirror")
        if held is None or held[1].flat_params.device != dev:
            import copy

            held = [None, copy.deepcopy(self.net).to(dev)]
            held[1].flat_params.requires_grad_(False)
        if held[0] != key:
            with torch.no_grad():
                held[1].flat_params.copy_(fp)          # (bumps the mirror's version: its weight image is re-packed)
                held[1].zstats.copy_(self.net.zstats)
            held[0] = key
        self.__dict__["_mirror"] = held
        return held[1]

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_mirror", None)                     # device scratch: never pickled / deep-copied
        return state

    def _kernel_log_prob(self, theta: Tensor, x: Tensor, want_noise: bool):
        net = self._kernel_net()
        if net is self.net:
            return self._raw_log_prob(net, theta, x, want_noise)
        dev = net.flat_params.device
        lp, noise = self._raw_log_prob(net, theta.to(dev), x.to(dev), want_noise)
        return lp.to(theta.device), (None if noise is None else noise.to(theta.device))

    def _kernel_sample(self, noise: Tensor, x: Tensor, want_ld: bool):
        net = self._kernel_net()
        if net is self.net:
            return self._raw_sample(net, noise, x, want_ld)
        dev = net.flat_params.device
        theta, ld = self._raw_sample(net, noise.to(dev), x.to(dev), want_ld)
        return theta.to(noise.device), (None if ld is None else ld.to(noise.device))

    def _autograd_log_prob(self, theta: Tensor, x: Tensor) -> Tensor:
        net = self._kernel_net()
        if net is self.net:
            return self._raw_autograd(net, theta, x, self.net.flat_params)
        # staged: `.to(dev)` is differentiable, so autograd itself carries d loss / d(theta, x, parameters) back to
        # the host tensors; the kernels read the mirror (same values as `flat`)
        dev = net.flat_params.device
        return self._raw_autograd(net, theta.to(dev), x.to(dev), self.net.flat_params.to(dev)).to(theta.device)

    # -- helpers ---------------------------------------------------------------------
    def _flatten_pair(self, input: Tensor, condition: Tensor) -> Tuple[Tensor, Tensor, int, int]:
        """(S,B,D)/(B,D) input + condition -> theta (S*B, D), x (x_rows, C), S, B without
        materialising a broadcast condition (the kernel indexes x[n % x_rows])."""
        self._check_input_shape(input)
        self._check_condition_shape(condition)
        input, S, B, cond_has_sample = self._broadcast_dims(input, condition)
        D = self.input_shape[0]
        theta = input.expand(S, B, D).reshape(S * B, D)
        emb = self._embed(condition)                      # (…, C): identity without an embedding net
        if cond_has_sample:
            x = emb.expand(S, B, self._cdim).reshape(S * B, self._cdim)
        else:
            x = emb.reshape(emb.shape[0], self._cdim)     # (1|B, C): n % x_rows does the broadcast
        if x.requires_grad and torch.is_grad_enabled() and x.shape[0] != S * B:
            # the kernels return d loss / d x per theta row: give every row its own condition row and let
            # autograd sum the copies
            x = x[torch.arange(S * B, device=x.device) % x.shape[0]]
        return theta.contiguous().float(), x.contiguous().float(), S, B

    # -- estimator surface -----------------------------------------------------------
    def log_prob(self, input: Tensor, condition: Tensor, **kwargs) -> Tensor:
        theta, x, S, B = self._flatten_pair(input, condition)
        needs_grad = torch.is_grad_enabled() and (theta.requires_grad or x.requires_grad
                                                  or self.net.flat_params.requires_grad)
        if needs_grad:
            lp = self._autograd_log_prob(theta, x)
        else:
            lp, _ = self._kernel_log_prob(theta, x, False)
        return lp.reshape(S, B)

    def loss(self, input: Tensor, condition: Tensor, **kwargs) -> Tensor:
        return -self.log_prob(input.unsqueeze(0), condition)[0]

    def inverse_transform(self, input: Tensor, condition: Tensor) -> Tensor:
        self._check_condition_shape(condition)
        bshape = torch.broadcast_shapes(input.shape[:-1], condition.shape[: condition.dim() - 1])
        theta = input.expand(bshape + (input.shape[-1],)).reshape(-1, input.shape[-1]).contiguous().float()
        with torch.no_grad():
            emb = self._embed(condition)
            x = emb.expand(bshape + (self._cdim,)).reshape(-1, self._cdim).contiguous().float()
            _, noise = self._kernel_log_prob(theta, x, True)
        return noise.reshape(bshape + (noise.shape[-1],))

    def sample_from_noise(self, noise: Tensor, condition: Tensor, with_logabsdet: bool = False):
        """theta = transform^{-1}(noise | condition); noise (N,D), condition (1|N, C)."""
        with torch.no_grad():
            emb = self._embed(condition)
            theta, ld = self._kernel_sample(noise.contiguous().float(),
                                            emb.reshape(emb.shape[0], self._cdim).contiguous().float(), with_logabsdet)
        return (theta, ld) if with_logabsdet else theta

    def sample(self, sample_shape: torch.Size, condition: Tensor, **kwargs) -> Tensor:
        self._check_condition_shape(condition)
        Bc = condition.shape[0]
        n = torch.Size(sample_shape).numel()
        D = self.input_shape[0]
        # nflows draws randn(Bc*n, D) viewed (Bc, n, D) and NFlowsFlow transposes to
        # (n, Bc, D) (nflows_flow.py:124-128): draw in that order, keep rows sample-major
        # so the kernel's x[row % Bc] pairs every draw with its condition.
        noise = torch.randn(Bc * n, D, device=condition.device, dtype=torch.float32)
        noise = noise.reshape(Bc, n, D).transpose(0, 1).reshape(n * Bc, D).contiguous()
        theta = self.sample_from_noise(noise, condition)
        return theta.reshape((*sample_shape, Bc, *self.input_shape))

    def sample_and_log_prob(self, sample_shape: torch.Size, condition: Tensor, **kwargs) -> Tuple[Tensor, Tensor]:
        self._check_condition_shape(condition)
        Bc = condition.shape[0]
        n = torch.Size(sample_shape).numel()
        D = self.input_shape[0]
        noise = torch.randn(Bc * n, D, device=condition.device, dtype=torch.float32)
        noise = noise.reshape(Bc, n, D).transpose(0, 1).reshape(n * Bc, D).contiguous()
        theta, ld = self.sample_from_noise(noise, condition, with_logabsdet=True)
        base = -0.5 * (noise**2).sum(1) - self.net._log_z.to(noise.dtype)
        logp = base - ld   # Flow.sample_and_log_prob: log p(noise) - logabsdet(inverse)
        return theta.reshape((*sample_shape, Bc, -1)), logp.reshape((*sample_shape, -1))
