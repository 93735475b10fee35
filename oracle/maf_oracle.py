"""CPU restatement (plain PyTorch, eager) of the `maf_rqs` density estimator sbi assembles in ``build_maf_rqs``
(sbi/neural_nets/net_builders/flow.py:212-330) -- TEST INFRASTRUCTURE, never the product path.

PARITY UNPINNED at the nflows boundary (nflows 0.14 is not importable here, oracle/__init__.py): the classes below
restate the *published* nflows algorithm op-for-op -- ``nflows.transforms.made`` (``MaskedLinear``,
``MaskedFeedforwardBlock``, ``MADE``), ``nflows.transforms.autoregressive``
(``MaskedPiecewiseRationalQuadraticAutoregressiveTransform``) and ``nflows.transforms.permutations``
(``RandomPermutation``) -- with the configuration sbi passes: ``use_residual_blocks=False, random_mask=False,
activation=tanh, tails="linear", tail_bound=3.0, num_blocks=2`` (flow.py:292-308).  The spline itself, the z-scoring
transform, the base density and the ``Flow`` / ``NFlowsFlow`` surface are the ones of oracle/nsf_oracle.py.
Module / attribute names follow nflows so that ``state_dict()`` keys match a real ``NFlowsFlow(build_maf_rqs(...))``.

Recalled details a real nflows install should confirm (tools/compare_with_nflows.py):
  * degrees: inputs 1..D; hidden `arange(H) % max(1, D-1) + min(1, D-1)`; outputs `repeat_interleave(1..D, P)`;
    hidden mask `deg_out >= deg_in`, output mask `deg_out > deg_in`;
  * MADE.forward: `t = initial(x); t += act(context_layer(c)); t = act(t)` (feed-forward blocks), then per block
    `t = act(masked_linear(t))` (the feed-forward block ignores the context), `final(t)`;
  * `MADE` has no `hidden_features` attribute, so the autoregressive transform's `if hasattr(net,
    "hidden_features")` branch does NOT divide the width / height logits by sqrt(H) (unlike the coupling transform
    with ResidualNet); `scale_by_sqrt_hidden=True` restates the other reading;
  * `RandomPermutation(features)`: buffer `_permutation = torch.randperm(features)`, forward `x[:, perm]`,
    inverse `x[:, argsort(perm)]`, logabsdet 0.
"""

from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch import Tensor, nn
from torch.nn import functional as F

from oracle.nsf_oracle import (CompositeTransform, Flow, NSFOracle, PointwiseAffineTransform, StandardNormal,
                               Standardize, sum_except_batch, unconstrained_rational_quadratic_spline,
                               z_standardization)


def _get_input_degrees(in_features: int) -> Tensor:
    return torch.arange(1, in_features + 1)


class MaskedLinear(nn.Linear):
    """nflows.transforms.made.MaskedLinear (random_mask=False)."""

    def __init__(self, in_degrees: Tensor, out_features: int, autoregressive_features: int, is_output: bool):
        super().__init__(in_features=len(in_degrees), out_features=out_features)
        if is_output:
            out_degrees = torch.repeat_interleave(_get_input_degrees(autoregressive_features),
                                                  out_features // autoregressive_features)
            mask = (out_degrees[..., None] > in_degrees).float()
        else:
            max_ = max(1, autoregressive_features - 1)
            min_ = min(1, autoregressive_features - 1)
            out_degrees = torch.arange(out_features) % max_ + min_
            mask = (out_degrees[..., None] >= in_degrees).float()
        self.register_buffer("mask", mask)
        self.register_buffer("degrees", out_degrees)

    def forward(self, x):
        return F.linear(x, self.weight * self.mask, self.bias)


class MaskedFeedforwardBlock(nn.Module):
    def __init__(self, in_degrees: Tensor, autoregressive_features: int, activation):
        super().__init__()
        self.linear = MaskedLinear(in_degrees, len(in_degrees), autoregressive_features, is_output=False)
        self.degrees = self.linear.degrees
        self.activation = activation

    def forward(self, inputs, context=None):
        return self.activation(self.linear(inputs))      # dropout p = 0; the context is not used by this block


class MADE(nn.Module):
    def __init__(self, features, hidden_features, context_features, num_blocks, output_multiplier,
                 activation=torch.tanh):
        super().__init__()
        self.initial_layer = MaskedLinear(_get_input_degrees(features), hidden_features, features, is_output=False)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, hidden_features)
        self.activation = activation
        blocks, prev = [], self.initial_layer.degrees
        for _ in range(num_blocks):
            blocks.append(MaskedFeedforwardBlock(prev, features, activation))
            prev = blocks[-1].degrees
        self.blocks = nn.ModuleList(blocks)
        self.final_layer = MaskedLinear(prev, features * output_multiplier, features, is_output=True)

    def forward(self, inputs, context=None):
        temps = self.initial_layer(inputs)
        if context is not None:
            temps = temps + self.activation(self.context_layer(context))
        temps = self.activation(temps)                   # use_residual_blocks=False
        for block in self.blocks:
            temps = block(temps, context)
        return self.final_layer(temps)


class MaskedPiecewiseRationalQuadraticAutoregressiveTransform(nn.Module):
    def __init__(self, features, hidden_features, context_features, num_bins=10, tail_bound=3.0, num_blocks=2,
                 min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, scale_by_sqrt_hidden=False):
        super().__init__()
        self.num_bins, self.tail_bound, self.features = num_bins, tail_bound, features
        self.min_bin_width, self.min_bin_height, self.min_derivative = min_bin_width, min_bin_height, min_derivative
        self.scale = float(np.sqrt(hidden_features)) if scale_by_sqrt_hidden else None
        self.autoregressive_net = MADE(features, hidden_features, context_features, num_blocks,
                                       output_multiplier=3 * num_bins - 1)

    def _elementwise(self, inputs, autoregressive_params, inverse):
        b, d = inputs.shape
        p = autoregressive_params.view(b, d, 3 * self.num_bins - 1)
        K = self.num_bins
        uw, uh, ud = p[..., :K], p[..., K : 2 * K], p[..., 2 * K :]
        if self.scale is not None:
            uw, uh = uw / self.scale, uh / self.scale
        out, ld = unconstrained_rational_quadratic_spline(
            inputs, uw, uh, ud, inverse=inverse, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
            min_bin_height=self.min_bin_height, min_derivative=self.min_derivative)
        return out, sum_except_batch(ld)

    def forward(self, inputs, context=None):
        return self._elementwise(inputs, self.autoregressive_net(inputs, context), inverse=False)

    def inverse(self, inputs, context=None):
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        for _ in range(inputs.shape[1]):                 # one more dimension becomes exact per pass
            params = self.autoregressive_net(outputs, context)
            outputs, logabsdet = self._elementwise(inputs, params, inverse=True)
        return outputs, logabsdet


class RandomPermutation(nn.Module):
    def __init__(self, features: int):
        super().__init__()
        self.register_buffer("_permutation", torch.randperm(features))

    def forward(self, inputs, context=None):
        return inputs[:, self._permutation], inputs.new_zeros(inputs.shape[0])

    def inverse(self, inputs, context=None):
        return inputs[:, torch.argsort(self._permutation)], inputs.new_zeros(inputs.shape[0])


class MAFRQSOracle(NSFOracle):
    """What ``NFlowsFlow(build_maf_rqs(batch_x=theta, batch_y=x, ...))`` computes; the NFlowsFlow surface
    (log_prob / loss / sample / inverse_transform / sample_from_noise) is inherited."""

    def __init__(self, batch_theta: Tensor, batch_x: Tensor, z_score_theta="independent", z_score_x="independent",
                 hidden_features=50, num_transforms=5, num_bins=10, tail_bound=3.0, num_blocks=2,
                 scale_by_sqrt_hidden=False):
        nn.Module.__init__(self)
        D, C = batch_theta[0].numel(), batch_x[0].numel()
        self.input_shape, self.condition_shape = batch_theta[0].shape, batch_x[0].shape
        transforms: List[nn.Module] = []
        for _ in range(num_transforms):                  # flow.py:290-311
            transforms.append(MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
                D, hidden_features, C, num_bins=num_bins, tail_bound=tail_bound, num_blocks=num_blocks,
                scale_by_sqrt_hidden=scale_by_sqrt_hidden))
            transforms.append(RandomPermutation(D))
        if z_score_theta in ("independent", "structured"):
            mean, std = z_standardization(batch_theta, z_score_theta == "structured", 1e-14)
            transforms = [PointwiseAffineTransform(shift=-mean / std, scale=1 / std)] + transforms
        if z_score_x in ("independent", "structured"):
            if len(batch_x) > 1:
                mean, std = z_standardization(batch_x, z_score_x == "structured", 1e-7)
            else:
                mean, std = torch.mean(batch_x, dim=0), torch.ones(1)
            embedding = nn.Sequential(Standardize(mean, std), nn.Identity())
        else:
            embedding = nn.Identity()
        dist = StandardNormal((D,))
        dist._log_z = dist._log_z.to(torch.float32)
        self.net = Flow(CompositeTransform(transforms), dist, embedding)
