"""CPU restatement (plain PyTorch, eager) of the NSF density estimator that
sbi assembles in ``build_nsf`` -- TEST INFRASTRUCTURE, never the product path.

PARITY UNPINNED (see oracle/__init__.py): nflows 0.14 is not importable here,
so every class below restates the *published* nflows algorithm op-for-op and
cites the sbi call site that pins its configuration.  Module/attribute names
follow nflows so that ``state_dict()`` keys match a real
``NFlowsFlow(build_nsf(...))`` and weights can be
exchanged with a real sbi install wherever one exists.

Reference call sites (all under /root/reference/sbi):
  * build_nsf wiring ........................ neural_nets/net_builders/flow.py:333-460
  * NFlowsFlow.log_prob/loss/sample ......... neural_nets/estimators/nflows_flow.py:77-151
  * theta z-score (PointwiseAffineTransform)  utils/sbiutils.py:226-247, 376-415
  * x z-score (Standardize) ................. utils/sbiutils.py:418-488
  * alternating masks ....................... utils/torchutils.py:396-410
  * searchsorted semantics (in-tree copy) ... utils/torchutils.py:449-463
  * base distribution ....................... neural_nets/net_builders/flow.py:1481-1488
nflows modules restated (third party, v0.14): transforms/base.py
(CompositeTransform), transforms/standard.py (PointwiseAffineTransform),
transforms/coupling.py, transforms/splines/rational_quadratic.py,
transforms/lu.py, nn/nets/resnet.py, flows/base.py, distributions/normal.py.
"""

from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn
from torch.nn import functional as F
from torch.nn import init

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


# --------------------------------------------------------------------------- utils
def searchsorted(bin_locations: Tensor, inputs: Tensor, eps: float = 1e-6) -> Tensor:
    """Same contract as sbi/utils/torchutils.py:449-463 (bumps the last knot in place)."""
    bin_locations[..., -1] += eps
    return torch.sum(inputs[..., None] >= bin_locations, dim=-1) - 1


def sum_except_batch(x: Tensor, num_batch_dims: int = 1) -> Tensor:
    """sbi/utils/torchutils.py:262-276."""
    return torch.sum(x, dim=list(range(num_batch_dims, x.ndimension())))


def repeat_rows(x: Tensor, num_reps: int) -> Tensor:
    """sbi/utils/torchutils.py:314-330."""
    shape = x.shape
    x = x.unsqueeze(1).expand(shape[0], num_reps, *shape[1:])
    return x.reshape(-1, *shape[1:])


def create_alternating_binary_mask(features: int, even: bool = True) -> Tensor:
    """sbi/utils/torchutils.py:396-410."""
    mask = torch.zeros(features).byte()
    mask[(0 if even else 1) :: 2] += 1
    return mask


# ------------------------------------------------------------------ spline (A.6/A.7)
def rational_quadratic_spline(
    inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives,
    inverse=False, left=0.0, right=1.0, bottom=0.0, top=1.0,
    min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
    min_derivative=DEFAULT_MIN_DERIVATIVE,
):
    """Monotone RQ spline of Durkan et al. 2019 as evaluated by nflows 0.14."""
    if torch.min(inputs) < left or torch.max(inputs) > right:
        raise ValueError("InputOutsideDomain")
    num_bins = unnormalized_widths.shape[-1]
    if min_bin_width * num_bins > 1.0 or min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin width/height too large for the number of bins")

    widths = F.softmax(unnormalized_widths, dim=-1)
    widths = min_bin_width + (1 - min_bin_width * num_bins) * widths
    cumwidths = torch.cumsum(widths, dim=-1)
    cumwidths = F.pad(cumwidths, pad=(1, 0), mode="constant", value=0.0)
    cumwidths = (right - left) * cumwidths + left
    cumwidths[..., 0] = left
    cumwidths[..., -1] = right
    widths = cumwidths[..., 1:] - cumwidths[..., :-1]

    derivatives = min_derivative + F.softplus(unnormalized_derivatives)

    heights = F.softmax(unnormalized_heights, dim=-1)
    heights = min_bin_height + (1 - min_bin_height * num_bins) * heights
    cumheights = torch.cumsum(heights, dim=-1)
    cumheights = F.pad(cumheights, pad=(1, 0), mode="constant", value=0.0)
    cumheights = (top - bottom) * cumheights + bottom
    cumheights[..., 0] = bottom
    cumheights[..., -1] = top
    heights = cumheights[..., 1:] - cumheights[..., :-1]

    if inverse:
        bin_idx = searchsorted(cumheights, inputs)[..., None]
    else:
        bin_idx = searchsorted(cumwidths, inputs)[..., None]

    input_cumwidths = cumwidths.gather(-1, bin_idx)[..., 0]
    input_bin_widths = widths.gather(-1, bin_idx)[..., 0]
    input_cumheights = cumheights.gather(-1, bin_idx)[..., 0]
    delta = heights / widths
    input_delta = delta.gather(-1, bin_idx)[..., 0]
    input_derivatives = derivatives.gather(-1, bin_idx)[..., 0]
    input_derivatives_plus_one = derivatives[..., 1:].gather(-1, bin_idx)[..., 0]
    input_heights = heights.gather(-1, bin_idx)[..., 0]

    if inverse:
        s = input_derivatives + input_derivatives_plus_one - 2 * input_delta
        a = (inputs - input_cumheights) * s + input_heights * (input_delta - input_derivatives)
        b = input_heights * input_derivatives - (inputs - input_cumheights) * s
        c = -input_delta * (inputs - input_cumheights)
        discriminant = b.pow(2) - 4 * a * c
        assert (discriminant >= 0).all()
        root = (2 * c) / (-b - torch.sqrt(discriminant))
        outputs = root * input_bin_widths + input_cumwidths
        theta_one_minus_theta = root * (1 - root)
        denominator = input_delta + s * theta_one_minus_theta
        derivative_numerator = input_delta.pow(2) * (
            input_derivatives_plus_one * root.pow(2)
            + 2 * input_delta * theta_one_minus_theta
            + input_derivatives * (1 - root).pow(2)
        )
        logabsdet = torch.log(derivative_numerator) - 2 * torch.log(denominator)
        return outputs, -logabsdet
    theta = (inputs - input_cumwidths) / input_bin_widths
    theta_one_minus_theta = theta * (1 - theta)
    numerator = input_heights * (input_delta * theta.pow(2) + input_derivatives * theta_one_minus_theta)
    denominator = input_delta + (
        (input_derivatives + input_derivatives_plus_one - 2 * input_delta) * theta_one_minus_theta
    )
    outputs = input_cumheights + numerator / denominator
    derivative_numerator = input_delta.pow(2) * (
        input_derivatives_plus_one * theta.pow(2)
        + 2 * input_delta * theta_one_minus_theta
        + input_derivatives * (1 - theta).pow(2)
    )
    logabsdet = torch.log(derivative_numerator) - 2 * torch.log(denominator)
    return outputs, logabsdet


def unconstrained_rational_quadratic_spline(
    inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives,
    inverse=False, tail_bound=1.0,
    min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
    min_derivative=DEFAULT_MIN_DERIVATIVE,
):
    """tails="linear" branch: identity (logabsdet 0) outside [-B, B]."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    outside = ~inside
    outputs = torch.zeros_like(inputs)
    logabsdet = torch.zeros_like(inputs)

    unnormalized_derivatives = F.pad(unnormalized_derivatives, pad=(1, 1))
    constant = np.log(np.exp(1 - min_derivative) - 1)
    unnormalized_derivatives[..., 0] = constant
    unnormalized_derivatives[..., -1] = constant

    outputs[outside] = inputs[outside]
    logabsdet[outside] = 0
    if torch.any(inside):
        outputs[inside], logabsdet[inside] = rational_quadratic_spline(
            inputs=inputs[inside],
            unnormalized_widths=unnormalized_widths[inside, :],
            unnormalized_heights=unnormalized_heights[inside, :],
            unnormalized_derivatives=unnormalized_derivatives[inside, :],
            inverse=inverse, left=-tail_bound, right=tail_bound,
            bottom=-tail_bound, top=tail_bound,
            min_bin_width=min_bin_width, min_bin_height=min_bin_height,
            min_derivative=min_derivative,
        )
    return outputs, logabsdet


# ------------------------------------------------------------- conditioner (A.7b)
class ResidualBlock(nn.Module):
    def __init__(self, features: int, context_features: Optional[int]):
        super().__init__()
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        self.linear_layers = nn.ModuleList([nn.Linear(features, features) for _ in range(2)])
        init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
        init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        temps = F.relu(inputs)
        temps = self.linear_layers[0](temps)
        temps = F.relu(temps)
        temps = self.linear_layers[1](temps)  # dropout p=0 is the identity
        if context is not None:
            temps = F.glu(torch.cat((temps, self.context_layer(context)), dim=1), dim=1)
        return inputs + temps


class ResidualNet(nn.Module):
    """nflows.nn.nets.ResidualNet with activation=relu, dropout 0, no batch norm
    (the configuration build_nsf uses, flow.py:411-419)."""

    def __init__(self, in_features, out_features, hidden_features, context_features=None, num_blocks=2):
        super().__init__()
        self.hidden_features = hidden_features
        self.context_features = context_features
        n_in = in_features + (context_features or 0)
        self.initial_layer = nn.Linear(n_in, hidden_features)
        self.blocks = nn.ModuleList(
            [ResidualBlock(hidden_features, context_features) for _ in range(num_blocks)]
        )
        self.final_layer = nn.Linear(hidden_features, out_features)

    def forward(self, inputs, context=None):
        if context is None:
            temps = self.initial_layer(inputs)
        else:
            temps = self.initial_layer(torch.cat((inputs, context), dim=1))
        for block in self.blocks:
            temps = block(temps, context=context)
        return self.final_layer(temps)


class ContextSplineMap(nn.Module):
    """sbi's conditioner for 1-D theta (flow.py:1419-1478): the spline parameters depend on the context
    only; `hidden_features` exists so the coupling transform applies its 1/sqrt(H) scaling.  Note the
    reference builds `[Linear, ReLU] * hidden_layers`, i.e. ONE hidden Linear reused hidden_layers times."""

    def __init__(self, in_features, out_features, hidden_features, context_features, hidden_layers=1):
        super().__init__()
        self.hidden_features = hidden_features
        layer_list = [nn.Linear(context_features, hidden_features), nn.ReLU()]
        layer_list += [nn.Linear(hidden_features, hidden_features), nn.ReLU()] * hidden_layers
        layer_list += [nn.Linear(hidden_features, out_features)]
        self.spline_predictor = nn.Sequential(*layer_list)

    def forward(self, inputs, context=None):
        return self.spline_predictor(context)


# --------------------------------------------------------------- transforms (A.2-A.8)
class PointwiseAffineTransform(nn.Module):
    def __init__(self, shift, scale):
        super().__init__()
        shift, scale = map(torch.as_tensor, (shift, scale))
        self.register_buffer("_shift", shift)
        self.register_buffer("_scale", scale)

    def forward(self, inputs, context=None):
        outputs = inputs * self._scale + self._shift
        log_scale = torch.log(torch.abs(self._scale))
        logabsdet = sum_except_batch(log_scale.expand(inputs.shape))
        return outputs, logabsdet

    def inverse(self, inputs, context=None):
        outputs = (inputs - self._shift) / self._scale
        log_scale = torch.log(torch.abs(self._scale))
        logabsdet = -sum_except_batch(log_scale.expand(inputs.shape))
        return outputs, logabsdet


class PiecewiseRationalQuadraticCouplingTransform(nn.Module):
    def __init__(self, mask, in_context, hidden_features, num_blocks, num_bins=10, tail_bound=3.0,
                 context_spline_map=False, hidden_layers_spline_context=1):
        super().__init__()
        mask = torch.as_tensor(mask)
        features_vector = torch.arange(len(mask))
        self.register_buffer("identity_features", features_vector.masked_select(mask <= 0))
        self.register_buffer("transform_features", features_vector.masked_select(mask > 0))
        self.num_bins = num_bins
        self.tail_bound = tail_bound
        if context_spline_map:
            self.transform_net = ContextSplineMap(
                len(self.identity_features), len(self.transform_features) * (3 * num_bins - 1), hidden_features,
                in_context, hidden_layers_spline_context)
        else:
            self.transform_net = ResidualNet(
                in_features=len(self.identity_features),
                out_features=len(self.transform_features) * (3 * num_bins - 1),
                hidden_features=hidden_features, context_features=in_context, num_blocks=num_blocks,
            )

    def _piecewise_cdf(self, inputs, transform_params, inverse):
        K = self.num_bins
        uw = transform_params[..., :K]
        uh = transform_params[..., K : 2 * K]
        ud = transform_params[..., 2 * K :]
        # nflows scales widths/heights (not derivatives) in place on the views.
        uw /= np.sqrt(self.transform_net.hidden_features)
        uh /= np.sqrt(self.transform_net.hidden_features)
        return unconstrained_rational_quadratic_spline(
            inputs, uw, uh, ud, inverse=inverse, tail_bound=self.tail_bound,
        )

    def _run(self, inputs, context, inverse):
        identity_split = inputs[:, self.identity_features]
        transform_split = inputs[:, self.transform_features]
        params = self.transform_net(identity_split, context)
        b, d = transform_split.shape
        transform_split, logabsdet = self._piecewise_cdf(transform_split, params.reshape(b, d, -1), inverse)
        logabsdet = sum_except_batch(logabsdet)
        outputs = torch.empty_like(inputs)
        outputs[:, self.identity_features] = identity_split
        outputs[:, self.transform_features] = transform_split
        return outputs, logabsdet

    def forward(self, inputs, context=None):
        return self._run(inputs, context, inverse=False)

    def inverse(self, inputs, context=None):
        return self._run(inputs, context, inverse=True)


class LULinear(nn.Module):
    def __init__(self, features: int, eps: float = 1e-3):
        super().__init__()
        self.features = features
        self.eps = eps
        self.bias = nn.Parameter(torch.zeros(features))
        self.lower_indices = np.tril_indices(features, k=-1)
        self.upper_indices = np.triu_indices(features, k=1)
        self.diag_indices = np.diag_indices(features)
        n_tri = ((features - 1) * features) // 2
        self.lower_entries = nn.Parameter(torch.zeros(n_tri))
        self.upper_entries = nn.Parameter(torch.zeros(n_tri))
        self.unconstrained_upper_diag = nn.Parameter(torch.zeros(features))
        init.constant_(self.unconstrained_upper_diag, float(np.log(np.exp(1 - eps) - 1)))

    @property
    def upper_diag(self):
        return F.softplus(self.unconstrained_upper_diag) + self.eps

    def _create_lower_upper(self):
        lower = self.lower_entries.new_zeros(self.features, self.features)
        lower[self.lower_indices[0], self.lower_indices[1]] = self.lower_entries
        lower[self.diag_indices[0], self.diag_indices[1]] = 1.0
        upper = self.upper_entries.new_zeros(self.features, self.features)
        upper[self.upper_indices[0], self.upper_indices[1]] = self.upper_entries
        upper[self.diag_indices[0], self.diag_indices[1]] = self.upper_diag
        return lower, upper

    def forward(self, inputs, context=None):
        lower, upper = self._create_lower_upper()
        outputs = F.linear(inputs, upper)
        outputs = F.linear(outputs, lower, self.bias)
        logabsdet = torch.sum(torch.log(self.upper_diag)) * inputs.new_ones(outputs.shape[0])
        return outputs, logabsdet

    def inverse(self, inputs, context=None):
        lower, upper = self._create_lower_upper()
        outputs = inputs - self.bias
        outputs = torch.linalg.solve_triangular(lower, outputs.t(), upper=False, unitriangular=True)
        outputs = torch.linalg.solve_triangular(upper, outputs, upper=True, unitriangular=False)
        outputs = outputs.t()
        logabsdet = -torch.sum(torch.log(self.upper_diag)) * inputs.new_ones(outputs.shape[0])
        return outputs, logabsdet


class CompositeTransform(nn.Module):
    def __init__(self, transforms: List[nn.Module]):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)

    def forward(self, inputs, context=None):
        total = inputs.new_zeros(inputs.shape[0])
        outputs = inputs
        for t in self._transforms:
            outputs, ld = t(outputs, context)
            total = total + ld
        return outputs, total

    def inverse(self, inputs, context=None):
        total = inputs.new_zeros(inputs.shape[0])
        outputs = inputs
        for t in reversed(list(self._transforms)):
            outputs, ld = t.inverse(outputs, context)
            total = total + ld
        return outputs, total


class StandardNormal(nn.Module):
    def __init__(self, shape):
        super().__init__()
        self._shape = torch.Size(shape)
        self.register_buffer(
            "_log_z", torch.tensor(0.5 * np.prod(shape) * np.log(2 * np.pi), dtype=torch.float64),
            persistent=False,
        )

    def log_prob(self, inputs):
        return -0.5 * sum_except_batch(inputs**2) - self._log_z

    def sample(self, num_samples, context_size):
        return torch.randn(context_size * num_samples, *self._shape, device=self._log_z.device)


class Standardize(nn.Module):
    """sbi/utils/sbiutils.py:418-428."""

    def __init__(self, mean, std):
        super().__init__()
        mean, std = map(torch.as_tensor, (mean, std))
        self.register_buffer("_mean", mean)
        self.register_buffer("_std", std)

    def forward(self, tensor):
        return (tensor - self._mean) / self._std


class Flow(nn.Module):
    def __init__(self, transform, distribution, embedding_net):
        super().__init__()
        self._transform = transform
        self._distribution = distribution
        self._embedding_net = embedding_net

    def log_prob(self, inputs, context):
        e = self._embedding_net(context)
        noise, logabsdet = self._transform(inputs, context=e)
        return self._distribution.log_prob(noise) + logabsdet

    def transform_to_noise(self, inputs, context):
        return self._transform(inputs, context=self._embedding_net(context))[0]

    def inverse_from_noise(self, noise, context):
        """transform.inverse for a GIVEN noise (the parity definition of `sample`)."""
        return self._transform.inverse(noise, context=self._embedding_net(context))

    def sample(self, num_samples, context):
        e = self._embedding_net(context)
        noise = self._distribution.sample(num_samples, context.shape[0])
        e = repeat_rows(e, num_samples)
        samples, _ = self._transform.inverse(noise, context=e)
        return samples.reshape(context.shape[0], num_samples, -1)


# ------------------------------------------------------------------- z-score stats
def z_standardization(batch_t: Tensor, structured: bool, min_std: float):
    """sbi/utils/sbiutils.py:376-415 (theta side, min_std 1e-14) and :431-488
    (x side, min_std 1e-7); NaN/Inf rows dropped first (handle_invalid_x)."""
    flat = batch_t.reshape(batch_t.shape[0], -1)
    valid = ~(torch.isnan(flat).any(1) | torch.isinf(flat).any(1))
    t = batch_t[valid]
    if structured:
        mean = torch.mean(t)
        sample_std = torch.std(t, dim=1)
        sample_std[sample_std < min_std] = min_std
        std = torch.mean(sample_std)
    else:
        mean = torch.mean(t, dim=0)
        std = torch.std(t, dim=0)
        std[std < min_std] = min_std
    return mean, std


# ----------------------------------------------------------------------- estimator
class NSFOracle(nn.Module):
    """What ``NFlowsFlow(build_nsf(batch_x=theta, batch_y=x, ...))`` computes.

    Shapes follow NFlowsFlow (nflows_flow.py:77-151): log_prob -> (S, B),
    loss -> (B,), sample -> (*shape, B, D).  ``net`` mirrors nflows' Flow.
    """

    def __init__(
        self, batch_theta: Tensor, batch_x: Tensor, z_score_theta="independent", z_score_x="independent",
        hidden_features=50, num_transforms=5, num_bins=10, tail_bound=3.0, num_blocks=2,
        hidden_layers_spline_context=1,
    ):
        super().__init__()
        D = batch_theta[0].numel()
        C = batch_x[0].numel()
        self.input_shape = batch_theta[0].shape
        self.condition_shape = batch_x[0].shape
        transforms: List[nn.Module] = []
        for i in range(num_transforms):
            if D == 1:   # flow.py:401-408, 426, 436: dummy mask [1], context-only conditioner, no LULinear
                transforms.append(
                    PiecewiseRationalQuadraticCouplingTransform(
                        torch.tensor([1], dtype=torch.uint8), C, hidden_features, num_blocks, num_bins=num_bins,
                        tail_bound=tail_bound, context_spline_map=True,
                        hidden_layers_spline_context=hidden_layers_spline_context))
                continue
            mask = create_alternating_binary_mask(D, even=(i % 2 == 0))
            transforms.append(
                PiecewiseRationalQuadraticCouplingTransform(
                    mask, C, hidden_features, num_blocks, num_bins=num_bins, tail_bound=tail_bound
                )
            )
            transforms.append(LULinear(D))
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
        dist._log_z = dist._log_z.to(torch.float32)  # flow.py:1486-1487
        self.net = Flow(CompositeTransform(transforms), dist, embedding)

    # -- NFlowsFlow surface -------------------------------------------------------
    def log_prob(self, input: Tensor, condition: Tensor) -> Tensor:
        if input.dim() <= len(self.input_shape) + 1:
            input = input.unsqueeze(0)
        S, Bi = input.shape[0], input.shape[1]
        has_s = condition.dim() > len(self.condition_shape) + 1
        Bc = condition.shape[1] if has_s else condition.shape[0]
        B = torch.broadcast_shapes((Bi,), (Bc,))[0]
        input = input.expand(S, B, *self.input_shape)
        if has_s:
            condition = condition.expand(S, B, *self.condition_shape)
        else:
            condition = condition.expand(B, *self.condition_shape).unsqueeze(0).expand(
                S, B, *self.condition_shape)
        lp = self.net.log_prob(input.reshape(S * B, -1), condition.reshape(S * B, *self.condition_shape))
        return lp.reshape(S, B)

    def loss(self, input: Tensor, condition: Tensor) -> Tensor:
        return -self.log_prob(input.unsqueeze(0), condition)[0]

    def sample(self, sample_shape, condition: Tensor) -> Tensor:
        n = torch.Size(sample_shape).numel()
        s = self.net.sample(n, condition).transpose(0, 1)
        return s.reshape((*sample_shape, condition.shape[0], *self.input_shape))

    def inverse_transform(self, input: Tensor, condition: Tensor) -> Tensor:
        bs = torch.broadcast_shapes(input.shape[:-1], condition.shape[: -len(self.condition_shape)])
        i = input.expand(bs + (input.shape[-1],)).reshape(-1, input.shape[-1])
        c = condition.expand(bs + self.condition_shape).reshape(-1, *self.condition_shape)
        return self.net.transform_to_noise(i, c).reshape(bs + (input.shape[-1],))

    def sample_from_noise(self, noise: Tensor, condition: Tensor) -> Tuple[Tensor, Tensor]:
        """theta = transform^{-1}(noise | condition) for explicit noise rows
        (noise (N, D), condition (N, C) or (1, C)).  Returns (theta, logabsdet)."""
        c = condition.expand(noise.shape[0], *self.condition_shape)
        return self.net.inverse_from_noise(noise, c)
