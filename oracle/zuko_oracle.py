"""CPU restatement (plain PyTorch, eager) of the `zuko_nsf` density estimator sbi assembles in ``build_zuko_nsf``
(sbi/neural_nets/net_builders/flow.py:578-640 -> build_zuko_flow :1082-1173 -> zuko.flows.NSF) behind
``ZukoFlow`` (sbi/neural_nets/estimators/zuko_flow.py:17-175) -- TEST INFRASTRUCTURE, never the product path.

PARITY UNPINNED: zuko (sbi requires >= 1.2.0, uv.lock pins 1.6.0) is not installed and not installable here, and the
reference's tests hold no numeric vectors for it.  The classes restate zuko's *published* algorithm as of the 1.x
series, op for op:
  * ``zuko.nn.MaskedLinear`` / ``MaskedMLP``: adjacency -> unique rows -> precedence matrix -> per-layer masks with
    hidden unit u taking the dependency pattern ``reachable[u % len(reachable)]``; ReLU between layers;
  * ``zuko.flows.autoregressive.MaskedAutoregressiveTransform``: adjacency ``order[:, None] > cat(order, -1 x C)``
    repeated 3K-1 times per feature, hyper-net on ``cat(x, c)``, output split (K, K, K-1); inverse by `passes`
    fixed-point sweeps from zeros; ``MAF``: orders arange / reversed alternating, base DiagNormal(0, 1);
  * ``zuko.transforms.MonotonicRQSTransform(widths, heights, derivatives, bound=5, slope=1e-3)``: soft-clipped logits
    ``w / (1 + |2 w / ln slope|)`` (derivatives ``d / (1 + |d / ln slope|)``), knots ``bound (2 cumsum(pad(softmax)) - 1)``,
    slopes ``exp(pad(d))``, bins by ``searchsorted(knots, x) - 1`` (identity outside), the rational-quadratic map of
    Durkan et al. and its closed-form inverse;
  * sbi's wiring: ``hidden_features = [hidden_features] * num_transforms`` (flow.py:1143-1144 -- the list length is
    the number of TRANSFORMS, so the default hyper-net has five hidden layers), z-scoring of theta as a leading
    affine transform (sbiutils.py:251-277), of x as ``Standardize`` in front (flow.py:1160-1161).
Anything a real zuko install contradicts is a bug in this file first, then in the kernels' host mirror.
"""

from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from oracle.nsf_oracle import Standardize, z_standardization


class MaskedLinear(nn.Linear):
    def __init__(self, adjacency: Tensor):
        super().__init__(adjacency.shape[1], adjacency.shape[0])
        self.register_buffer("mask", adjacency.bool())

    def forward(self, x):
        return F.linear(x, self.mask * self.weight, self.bias)


def masked_mlp_masks(adjacency: Tensor, hidden_features: Sequence[int]) -> List[Tensor]:
    """The per-layer boolean masks zuko.nn.MaskedMLP builds from an (out x in) adjacency matrix."""
    adjacency = adjacency.bool()
    out_features = adjacency.shape[0]
    uniq, inverse = torch.unique(adjacency, dim=0, return_inverse=True)
    # P_ij = 1 if A_ik = 1 for all k such that A_jk = 1
    precedence = uniq.double() @ uniq.double().t() == uniq.double().sum(dim=-1)
    masks, indices = [], None
    for i, features in enumerate((*hidden_features, out_features)):
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


class MaskedMLP(nn.Sequential):
    def __init__(self, adjacency: Tensor, hidden_features: Sequence[int]):
        layers: List[nn.Module] = []
        for m in masked_mlp_masks(adjacency, hidden_features):
            layers += [MaskedLinear(m), nn.ReLU()]
        super().__init__(*layers[:-1])


LOG_SLOPE = math.log(1e-3)


def rqs_params(widths, heights, derivatives, bound=5.0):
    widths = widths / (1 + abs(2 * widths / LOG_SLOPE))
    heights = heights / (1 + abs(2 * heights / LOG_SLOPE))
    derivatives = derivatives / (1 + abs(derivatives / LOG_SLOPE))
    widths = F.pad(F.softmax(widths, dim=-1), (1, 0), value=0)
    heights = F.pad(F.softmax(heights, dim=-1), (1, 0), value=0)
    derivatives = F.pad(derivatives, (1, 1), value=0)
    horizontal = bound * (2 * torch.cumsum(widths, dim=-1) - 1)
    vertical = bound * (2 * torch.cumsum(heights, dim=-1) - 1)
    return horizontal, vertical, torch.exp(derivatives)


def _bin(horizontal, vertical, derivatives, k):
    bins = horizontal.shape[-1] - 1
    mask = torch.logical_and(0 <= k, k < bins)
    k = k % bins
    k0, k1 = k[..., None], k[..., None] + 1
    g = lambda t, i: t.gather(-1, i).squeeze(-1)   # noqa: E731
    x0, x1 = g(horizontal, k0), g(horizontal, k1)
    y0, y1 = g(vertical, k0), g(vertical, k1)
    d0, d1 = g(derivatives, k0), g(derivatives, k1)
    s = (y1 - y0) / (x1 - x0)
    return mask, x0, x1, y0, y1, d0, d1, s


def rqs_forward(x, horizontal, vertical, derivatives):
    k = torch.searchsorted(horizontal, x[..., None]).squeeze(-1) - 1
    mask, x0, x1, y0, y1, d0, d1, s = _bin(horizontal, vertical, derivatives, k)
    z = mask * (x - x0) / (x1 - x0)
    y = y0 + (y1 - y0) * (s * z**2 + d0 * z * (1 - z)) / (s + (d0 + d1 - 2 * s) * z * (1 - z))
    jac = s**2 * (2 * s * z * (1 - z) + d0 * (1 - z) ** 2 + d1 * z**2) / (s + (d0 + d1 - 2 * s) * z * (1 - z)) ** 2
    return torch.where(mask, y, x), torch.log(jac) * mask


def rqs_inverse(y, horizontal, vertical, derivatives):
    k = torch.searchsorted(vertical, y[..., None]).squeeze(-1) - 1
    mask, x0, x1, y0, y1, d0, d1, s = _bin(horizontal, vertical, derivatives, k)
    y_ = mask * (y - y0)
    a = (y1 - y0) * (s - d0) + y_ * (d0 + d1 - 2 * s)
    b = (y1 - y0) * d0 - y_ * (d0 + d1 - 2 * s)
    c = -s * y_
    z = 2 * c / (-b - (b**2 - 4 * a * c).sqrt())
    x = x0 + z * (x1 - x0)
    return torch.where(mask, x, y)


class MaskedAutoregressiveTransform(nn.Module):
    def __init__(self, features: int, context: int, order: Tensor, hidden_features: Sequence[int], bins: int,
                 bound: float = 5.0):
        super().__init__()
        self.features, self.bins, self.bound = features, bins, bound
        self.total = 3 * bins - 1
        self.register_buffer("order", order.clone(), persistent=False)
        in_order = torch.cat((order, torch.full((context,), -1, dtype=order.dtype)))
        out_order = torch.repeat_interleave(order, self.total)
        self.hyper = MaskedMLP(out_order[:, None] > in_order, hidden_features)

    def _params(self, x, c):
        phi = self.hyper(torch.cat((x, c), dim=-1))
        phi = phi.unflatten(-1, (self.features, self.total))
        K = self.bins
        return rqs_params(phi[..., :K], phi[..., K : 2 * K], phi[..., 2 * K :], self.bound)

    def forward(self, x, c):
        y, ladj = rqs_forward(x, *self._params(x, c))
        return y, ladj.sum(-1)

    def inverse(self, y, c):
        x = torch.zeros_like(y)
        for _ in range(self.features):           # passes = features: fully autoregressive
            x = rqs_inverse(y, *self._params(x, c))
        return x, -self.forward(x, c)[1]


class AffineZ(nn.Module):
    """zuko.flows.UnconditionalTransform(AffineTransform, loc, scale, buffer=True): y = loc + scale x"""

    def __init__(self, loc, scale):
        super().__init__()
        self.register_buffer("loc", torch.as_tensor(loc))
        self.register_buffer("scale", torch.as_tensor(scale))

    def forward(self, x, c=None):
        return self.loc + self.scale * x, torch.log(torch.abs(self.scale)).expand(x.shape).sum(-1)

    def inverse(self, y, c=None):
        return (y - self.loc) / self.scale, -torch.log(torch.abs(self.scale)).expand(y.shape).sum(-1)


class ZukoNSFOracle(nn.Module):
    """What ``ZukoFlow(build_zuko_nsf(batch_x=theta, batch_y=x, ...))`` computes: log_prob -> (S, B), loss -> (B,),
    sample -> (*shape, B, D), plus the transform for given noise."""

    def __init__(self, batch_theta: Tensor, batch_x: Tensor, z_score_theta="independent", z_score_x="independent",
                 hidden_features=50, num_transforms=5, num_bins=10):
        super().__init__()
        D, C = batch_theta[0].numel(), batch_x[0].numel()
        self.input_shape, self.condition_shape = batch_theta[0].shape, batch_x[0].shape
        hidden = [hidden_features] * num_transforms if isinstance(hidden_features, int) else list(hidden_features)
        orders = [torch.arange(D), torch.flipud(torch.arange(D))]
        ts: List[nn.Module] = [MaskedAutoregressiveTransform(D, C, orders[i % 2], hidden, num_bins)
                               for i in range(num_transforms)]
        if z_score_theta in ("independent", "structured"):
            mean, std = z_standardization(batch_theta, z_score_theta == "structured", 1e-14)
            ts = [AffineZ(-mean / std, 1 / std)] + ts
        self.transforms = nn.ModuleList(ts)
        if z_score_x in ("independent", "structured"):
            if len(batch_x) > 1:
                mean, std = z_standardization(batch_x, z_score_x == "structured", 1e-7)
            else:
                mean, std = torch.mean(batch_x, dim=0), torch.ones(1)
            self.embedding = nn.Sequential(Standardize(mean, std), nn.Identity())
        else:
            self.embedding = nn.Identity()
        self.D = D

    def _forward(self, theta, c):
        total = theta.new_zeros(theta.shape[0])
        z = theta
        for t in self.transforms:
            z, ld = t(z, c)
            total = total + ld
        return z, total

    def _log_prob_flat(self, theta, x):
        z, ld = self._forward(theta, self.embedding(x))
        return (-0.5 * z**2 - 0.5 * math.log(2 * math.pi)).sum(-1) + ld

    def log_prob(self, input: Tensor, condition: Tensor) -> Tensor:
        if input.dim() <= len(self.input_shape) + 1:
            input = input.unsqueeze(0)
        S, Bi = input.shape[0], input.shape[1]
        has_s = condition.dim() > len(self.condition_shape) + 1
        Bc = condition.shape[1] if has_s else condition.shape[0]
        B = torch.broadcast_shapes((Bi,), (Bc,))[0]
        input = input.expand(S, B, *self.input_shape)
        condition = condition.expand(S, B, *self.condition_shape) if has_s else \
            condition.expand(B, *self.condition_shape).unsqueeze(0).expand(S, B, *self.condition_shape)
        return self._log_prob_flat(input.reshape(S * B, -1), condition.reshape(S * B, -1)).reshape(S, B)

    def loss(self, input: Tensor, condition: Tensor) -> Tensor:
        return -self.log_prob(input.unsqueeze(0), condition)[0]

    def inverse_transform(self, input: Tensor, condition: Tensor) -> Tensor:
        c = condition.expand(input.shape[0], *self.condition_shape)
        return self._forward(input, self.embedding(c))[0]

    def sample_from_noise(self, noise: Tensor, condition: Tensor) -> Tuple[Tensor, Tensor]:
        c = self.embedding(condition.expand(noise.shape[0], *self.condition_shape))
        total = noise.new_zeros(noise.shape[0])
        x = noise
        for t in reversed(list(self.transforms)):
            x, ld = t.inverse(x, c)
            total = total + ld
        return x, total

    def sample(self, sample_shape, condition: Tensor) -> Tensor:
        n = torch.Size(sample_shape).numel()
        B = condition.shape[0]
        noise = torch.randn(n, B, self.D)            # DiagNormal.rsample((n,)) with batch shape (B,)
        c = condition.unsqueeze(0).expand(n, B, *self.condition_shape).reshape(n * B, -1)
        x, _ = self.sample_from_noise(noise.reshape(n * B, self.D), c)
        return x.reshape((*sample_shape, B, self.D))
