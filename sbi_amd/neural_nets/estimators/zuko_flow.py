"""`zuko_nsf` density estimator on the MI355X HIP kernels.

``ZukoNSFFlow`` is the drop-in for sbi's ``ZukoFlow(build_zuko_nsf(...))`` (sbi/neural_nets/net_builders/flow.py:578-640,
:1082-1173; estimators/zuko_flow.py:17-175): zuko's autoregressive neural spline flow -- per transform a masked MLP
hyper-net on ``[theta ; embedded x]`` (ReLU, adjacency masks, autoregressive order alternating arange / reversed)
producing the 3K-1 parameters of a ``MonotonicRQSTransform`` (bound 5) for every dimension; z-scoring of theta as a
leading affine transform, base N(0, I).  The arithmetic runs on the maf kernels of ``libsbi_amd_nsf.so`` in their
``variant = 1`` configuration (include/sbi_amd_maf.h); the adjacency masks are computed here, on the host, by zuko's
``MaskedMLP`` rule and handed to the kernels as a 0/1 buffer in the layout of the flat parameters.  No CPU fallback.

zuko is not installable in the build container: layout / key names follow zuko 1.x as restated in
oracle/zuko_oracle.py (parity unpinned at that boundary; the loader is keyed on zuko's ``hyper.{2l}`` names).
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
from sbi_amd.neural_nets.estimators.maf_flow import MAFNet, MAFRQSFlow


def masked_mlp_masks(adjacency: Tensor, hidden_features) -> List[Tensor]:
    """Per-layer boolean masks of zuko.nn.MaskedMLP for an (out x in) adjacency matrix: rows with equal dependencies
    are merged, ``precedence[p, q]`` says pattern q's dependencies are contained in p's, and hidden unit u of every
    layer takes pattern ``reachable[u % len(reachable)]``."""
    adjacency = adjacency.bool()
    uniq, inverse = torch.unique(adjacency, dim=0, return_inverse=True)
    precedence = uniq.double() @ uniq.double().t() == uniq.double().sum(dim=-1)
    masks, indices = [], None
    for i, features in enumerate((*hidden_features, adjacency.shape[0])):
        mask = precedence[:, indices] if i > 0 else uniq
        if (~mask).all():
            raise ValueError("The adjacency matrix leads to a null Jacobian.")
        if i < len(hidden_features):
            reachable = mask.sum(dim=-1).nonzero().squeeze(dim=-1)
            indices = reachable[torch.arange(features) % len(reachable)]
            mask = mask[indices]
        else:
            mask = mask[inverse]
        masks.append(mask)
    return masks


@dataclass(frozen=True)
class ZukoHyper:
    """What ``build_zuko_nsf`` bakes into ``zuko.flows.NSF`` (flow.py:578-640, 1143-1152)."""

    D: int
    C: int
    hidden_features: int = 50
    num_transforms: int = 5
    num_bins: int = 10
    num_hidden_layers: int = 5          # sbi: [hidden_features] * num_transforms
    tail_bound: float = 5.0

    @property
    def num_blocks(self) -> int:        # hidden -> hidden linears
        return self.num_hidden_layers - 1

    def c_config(self) -> _lib.MAFConfigC:
        return _lib.MAFConfigC(self.D, self.C, self.hidden_features, self.num_bins, self.num_transforms,
                               self.num_blocks, self.tail_bound, 0.0, 0.0, 0.0, 0, 1)

    def layer_entries(self) -> List[Tuple[str, Tuple[int, ...], int]]:
        """(zuko sub-key, shape, layer index | -1 for biases) in flat order for one transform."""
        H, D, C, P = self.hidden_features, self.D, self.C, 3 * self.num_bins - 1
        out = [("hyper.0.weight", (H, D + C), 0), ("hyper.0.bias", (H,), -1)]
        for l in range(1, self.num_hidden_layers):
            out += [(f"hyper.{2 * l}.weight", (H, H), l), (f"hyper.{2 * l}.bias", (H,), -1)]
        L = self.num_hidden_layers
        out += [(f"hyper.{2 * L}.weight", (D * P, H), L), (f"hyper.{2 * L}.bias", (D * P,), -1)]
        return out

    def layer_params(self) -> int:
        return sum(int(np.prod(s)) for _, s, _ in self.layer_entries())

    def param_count(self) -> int:
        return self.num_transforms * self.layer_params()

    def order(self, t: int) -> Tensor:
        """zuko.flows.MAF: orders alternate arange / reversed (randperm=False)."""
        o = torch.arange(self.D)
        return o if t % 2 == 0 else torch.flipud(o)

    def masks(self, t: int) -> List[Tensor]:
        order = self.order(t)
        in_order = torch.cat((order, torch.full((self.C,), -1, dtype=order.dtype)))
        out_order = torch.repeat_interleave(order, 3 * self.num_bins - 1)
        return masked_mlp_masks(out_order[:, None] > in_order, [self.hidden_features] * self.num_hidden_layers)


class ZukoNSFNet(MAFNet):
    """Parameter / buffer holder in the role of ``zuko.flows.Flow`` for zuko_nsf.  ``perms[t]`` holds transform t's
    autoregressive order; ``mask_flat`` the adjacency masks in the layout of ``flat_params`` (bias slots 1)."""

    def __init__(self, hyper: ZukoHyper, zstats: Tensor, z_score_theta: bool, z_score_x: bool,
                 dtype: torch.dtype = torch.float32):
        nn.Module.__init__(self)
        self.hyper = hyper
        self.z_score_theta = z_score_theta
        self.z_score_x = z_score_x
        self.flat_params = nn.Parameter(torch.zeros(hyper.param_count(), dtype=torch.float32))
        self.register_buffer("zstats", zstats.to(torch.float32).contiguous())
        self.register_buffer("perms", torch.stack([hyper.order(t) for t in range(hyper.num_transforms)]).to(torch.int32))
        chunks = []
        for t in range(hyper.num_transforms):
            masks = hyper.masks(t)
            for (_key, shape, layer) in hyper.layer_entries():
                chunks.append(masks[layer].float().reshape(-1) if layer >= 0 else torch.ones(shape))
        self.register_buffer("mask_flat", torch.cat(chunks).contiguous())
        self.register_buffer("_log_z", torch.tensor(0.5 * hyper.D * math.log(2 * math.pi),
                                                    dtype=torch.float64).to(dtype), persistent=False)
        self.reset_parameters()

    def kernel_masks(self) -> Optional[Tensor]:
        return self.mask_flat

    @torch.no_grad()
    def reset_parameters(self) -> None:
        """zuko's construction order: per transform the MaskedLinear layers of the hyper-net first to last, each
        with nn.Linear's default initialisation (no permutation is drawn: randperm=False)."""
        h = self.hyper
        chunks: List[Tensor] = []
        for _t in range(h.num_transforms):
            for key, shape, layer in h.layer_entries():
                if layer >= 0:
                    m = nn.Linear(shape[1], shape[0])
                    chunks += [m.weight.detach().reshape(-1), m.bias.detach().reshape(-1)]
        flat = torch.cat(chunks)
        assert flat.numel() == self.flat_params.numel()
        self.flat_params.copy_(flat)

    def _slices(self):
        h = self.hyper
        first = 1 if self.z_score_theta else 0
        off = 0
        for t in range(h.num_transforms):
            pre = f"transforms.{first + t}."
            for key, shape, _layer in h.layer_entries():
                n = int(np.prod(shape))
                yield pre + key, off, n, shape
                off += n

    def zuko_state_dict(self, prefix: str = "") -> "OrderedDict[str, Tensor]":
        """Weights under the key names of oracle/zuko_oracle.py (= zuko's ``transform.transforms.{i}.hyper.{2l}``
        naming up to the container prefix)."""
        h = self.hyper
        sd: "OrderedDict[str, Tensor]" = OrderedDict()
        flat = self.flat_params.detach()
        if self.z_score_theta:
            sd[prefix + "transforms.0.loc"] = self.zstats[: h.D].clone()
            sd[prefix + "transforms.0.scale"] = self.zstats[h.D : 2 * h.D].clone()
        for key, off, n, shape in self._slices():
            sd[prefix + key] = flat[off : off + n].reshape(shape).clone()
        if self.z_score_x:
            sd[prefix + "embedding.0._mean"] = self.zstats[2 * h.D : 2 * h.D + h.C].clone()
            sd[prefix + "embedding.0._std"] = self.zstats[2 * h.D + h.C :].clone()
        return sd

    @torch.no_grad()
    def load_zuko_state_dict(self, sd: Dict[str, Tensor], prefix: str = "") -> None:
        h = self.hyper
        for key, off, n, shape in self._slices():
            src = sd[prefix + key]
            if tuple(src.shape) != tuple(shape):
                raise ValueError(f"{key}: expected {shape}, got {tuple(src.shape)}")
            self.flat_params[off : off + n].copy_(src.reshape(-1).to(self.flat_params))
            mkey = prefix + key.replace(".weight", ".mask")
            if key.endswith(".weight") and mkey in sd:      # a real zuko state dict carries its masks: they must agree
                if not torch.equal(sd[mkey].bool().cpu(), self.mask_flat[off : off + n].reshape(shape).bool().cpu()):
                    raise ValueError(f"{mkey}: adjacency mask differs from the one this build derives")
        if self.z_score_theta:
            self.zstats[: h.D].copy_(sd[prefix + "transforms.0.loc"].reshape(-1).expand(h.D))
            self.zstats[h.D : 2 * h.D].copy_(sd[prefix + "transforms.0.scale"].reshape(-1).expand(h.D))
        if self.z_score_x:
            self.zstats[2 * h.D : 2 * h.D + h.C].copy_(sd[prefix + "embedding.0._mean"].reshape(-1).expand(h.C))
            self.zstats[2 * h.D + h.C :].copy_(sd[prefix + "embedding.0._std"].reshape(-1).expand(h.C))
        self.__dict__.pop("_packed_cache", None)

    # the nflows-named exchange of the parent class does not apply
    def nflows_state_dict(self, prefix: str = "net."):
        raise NotImplementedError("zuko_nsf exchanges weights under zuko's key names: zuko_state_dict()")

    def load_nflows_state_dict(self, sd, prefix: str = "net."):
        raise NotImplementedError("zuko_nsf exchanges weights under zuko's key names: load_zuko_state_dict()")


class ZukoNSFFlow(MAFRQSFlow):
    r"""zuko's autoregressive neural spline flow :math:`p(\theta|x)` evaluated by the gfx950 kernels."""

    def _draw_noise(self, n: int, Bc: int, device) -> Tensor:
        """zuko draws the base noise as ``DiagNormal.rsample((n,))`` with batch shape (Bc,): an (n, Bc, D) normal
        tensor, already sample-major (nflows draws (Bc * n, D) and transposes)."""
        return torch.randn(n * Bc, self.input_shape[0], device=device, dtype=torch.float32)

    def sample(self, sample_shape: torch.Size, condition: Tensor, **kwargs) -> Tensor:
        self._check_condition_shape(condition)
        Bc, n = condition.shape[0], torch.Size(sample_shape).numel()
        theta = self.sample_from_noise(self._draw_noise(n, Bc, condition.device), condition)
        return theta.reshape((*sample_shape, Bc, *self.input_shape))

    def sample_and_log_prob(self, sample_shape: torch.Size, condition: Tensor, **kwargs):
        self._check_condition_shape(condition)
        Bc, n = condition.shape[0], torch.Size(sample_shape).numel()
        noise = self._draw_noise(n, Bc, condition.device)
        theta, ld = self.sample_from_noise(noise, condition, with_logabsdet=True)
        base = -0.5 * (noise**2).sum(1) - self.net._log_z.to(noise.dtype)
        return theta.reshape((*sample_shape, Bc, -1)), (base - ld).reshape((*sample_shape, -1))
