"""Flow-matching vector-field estimator running on the HIP kernels of csrc/fmpe.hip.

Host-side mirror of sbi's default FMPE estimator (same method names, argument meaning and shapes):
  FlowMatchingEstimator      sbi/neural_nets/estimators/flowmatching_estimator.py:15-372
  VectorFieldMLP             sbi/neural_nets/net_builders/vector_field_nets.py:610-719
  ConditionalVectorFieldEstimator (t_min / t_max / mean_base / std_base / solve_schedule)
                             sbi/neural_nets/estimators/base.py:309-520

Supported: the default configuration family -- ``net="mlp"`` with GELU, LayerNorm, skip connections and the
sinusoidal time embedding, flat theta and flat (embedded) x, no Gaussian baseline, no composed standardisation.
Anything else raises; there is no PyTorch fallback for the network.
"""

from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from sbi_amd import _lib
from sbi_amd.neural_nets.estimators.base import ConditionalEstimator


@dataclass(frozen=True)
class FMPEHyper:
    D: int
    C: int
    hidden_features: int = 100
    num_layers: int = 5
    time_embedding_dim: int = 32
    sinusoidal_max_freq: float = 1000.0
    noise_scale: float = 1e-3
    ln_eps: float = 1e-5

    def c_config(self) -> _lib.FMPEConfigC:
        return _lib.FMPEConfigC(self.D, self.C, self.hidden_features, self.num_layers, self.time_embedding_dim,
                                self.sinusoidal_max_freq, self.noise_scale, self.ln_eps)

    def entries(self):
        """(reference state-dict key below ``net.``, shape) in flat-buffer order (include/sbi_amd_fmpe.h)."""
        H, D, C, E, L = self.hidden_features, self.D, self.C, self.time_embedding_dim, self.num_layers
        out = [("input_layer.weight", (H, D)), ("input_layer.bias", (H,)),
               ("condition_layer.weight", (H, C)), ("condition_layer.bias", (H,)),
               ("input_merge_layer.weight", (H, 2 * H)), ("input_merge_layer.bias", (H,)),
               ("time_linear_layer.weight", (H, E)), ("time_linear_layer.bias", (H,))]
        for l in range(L):
            out += [(f"layers.{l}.weight", (H, H)), (f"layers.{l}.bias", (H,))]
        for l in range(L):
            out += [(f"layers_norm.{l}.weight", (H,)), (f"layers_norm.{l}.bias", (H,))]
        out += [("output_layer.weight", (D, H)), ("output_layer.bias", (D,))]
        return out

    def param_count(self) -> int:
        return sum(math.prod(s) for _, s in self.entries())


class VectorFieldMLPParams(nn.Module):
    """Flat parameter buffer + z-scoring statistics: the role VectorFieldMLP's modules play in sbi.

    ``zstats`` = [mean_0 (D), std_0 (D), x mean (C), x std (C)].
    """

    def __init__(self, hyper: FMPEHyper, zstats: Tensor):
        super().__init__()
        self.hyper = hyper
        self.flat_params = nn.Parameter(torch.zeros(hyper.param_count(), dtype=torch.float32))
        self.register_buffer("zstats", zstats.to(torch.float32).contiguous())
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self) -> None:
        """torch's nn.Linear / nn.LayerNorm defaults, drawn in the order VectorFieldMLP.__init__ builds its
        modules (vector_field_nets.py:648-681); output_layer.weight is zero (:681)."""
        h = self.hyper
        H, D, C, E, L = h.hidden_features, h.D, h.C, h.time_embedding_dim, h.num_layers
        mods = OrderedDict()
        mods["input_layer"] = nn.Linear(D, H)
        mods["condition_layer"] = nn.Linear(C, H)
        mods["input_merge_layer"] = nn.Linear(2 * H, H)
        for l in range(L):
            mods[f"layers.{l}"] = nn.Linear(H, H)
        for l in range(L):
            mods[f"layers_norm.{l}"] = nn.LayerNorm(H)
        mods["time_linear_layer"] = nn.Linear(E, H)
        mods["output_layer"] = nn.Linear(H, D)
        nn.init.zeros_(mods["output_layer"].weight)
        chunks = []
        for key, _ in h.entries():
            mod, attr = key.rsplit(".", 1)
            chunks.append(getattr(mods[mod], attr).detach().reshape(-1))
        self.flat_params.copy_(torch.cat(chunks))

    def slices(self):
        off = 0
        for key, shape in self.hyper.entries():
            n = math.prod(shape)
            yield key, off, n, shape
            off += n

    def reference_state_dict(self) -> "OrderedDict[str, Tensor]":
        """Keys of sbi's ``FlowMatchingEstimator.state_dict()`` (the ones this path owns)."""
        h = self.hyper
        sd: "OrderedDict[str, Tensor]" = OrderedDict()
        flat = self.flat_params.detach()
        for key, off, n, shape in self.slices():
            sd["net." + key] = flat[off : off + n].reshape(shape).clone()
        sd["mean_0"] = self.zstats[: h.D].clone()
        sd["std_0"] = self.zstats[h.D : 2 * h.D].clone()
        sd["_embedding_net.0._mean"] = self.zstats[2 * h.D : 2 * h.D + h.C].clone()
        sd["_embedding_net.0._std"] = self.zstats[2 * h.D + h.C :].clone()
        return sd

    @torch.no_grad()
    def load_reference_state_dict(self, sd: Dict[str, Tensor]) -> None:
        h = self.hyper
        for key, off, n, shape in self.slices():
            src = sd["net." + key]
            if tuple(src.shape) != tuple(shape):
                raise ValueError(f"{key}: expected {shape}, got {tuple(src.shape)}")
            self.flat_params[off : off + n].copy_(src.reshape(-1).to(self.flat_params))
        self.zstats[: h.D].copy_(sd["mean_0"].reshape(-1))
        self.zstats[h.D : 2 * h.D].copy_(sd["std_0"].reshape(-1))
        if "_embedding_net.0._mean" in sd:
            self.zstats[2 * h.D : 2 * h.D + h.C].copy_(sd["_embedding_net.0._mean"].reshape(-1).expand(h.C))
            self.zstats[2 * h.D + h.C :].copy_(sd["_embedding_net.0._std"].reshape(-1).expand(h.C))


# --------------------------------------------------------------------- kernel calls
def packed_weights(net: VectorFieldMLPParams) -> Tensor:
    fp = net.flat_params
    key = (fp.data_ptr(), fp._version, str(fp.device))
    cache = net.__dict__.get("_packed_cache")
    if cache is not None and cache[0] == key:
        return cache[1]
    dev = _lib.require_device(fp)
    lib = _lib.load()
    cfg = net.hyper.c_config()
    n = lib.sbi_amd_fmpe_packed_floats(cfg)
    if n < 0:
        _lib.check(int(n), "fmpe_packed_floats")
    packed = cache[1] if (cache is not None and cache[1].device == dev and cache[1].numel() == n) else \
        torch.zeros(int(n), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_fmpe_pack(cfg, _lib.ptr(fp), _lib.ptr(packed), _lib.current_stream(dev))
    _lib.check(rc, "fmpe_pack")
    net.__dict__["_packed_cache"] = (key, packed)
    return packed


def velocity(net: VectorFieldMLPParams, theta_t: Tensor, x: Tensor, times: Tensor) -> Tensor:
    """theta_t (n, D), x (n, C) or (1, C), times (n,) or (1,) -> velocity (n, D)."""
    dev = _lib.require_device(theta_t, x, times, net.flat_params)
    n = theta_t.shape[0]
    out = torch.empty_like(theta_t)
    if n == 0:
        return out
    with torch.cuda.device(dev):
        rc = _lib.load().sbi_amd_fmpe_velocity(
            net.hyper.c_config(), _lib.ptr(packed_weights(net)), _lib.ptr(net.zstats), _lib.ptr(theta_t),
            _lib.ptr(x), x.shape[0], _lib.ptr(times), times.numel(), n, _lib.ptr(out), _lib.current_stream(dev))
    _lib.check(rc, "fmpe_velocity")
    return out


def velocity_and_divergence(net: VectorFieldMLPParams, theta_t: Tensor, x: Tensor, times: Tensor,
                            v_out: Optional[Tensor] = None, div_out: Optional[Tensor] = None):
    """theta_t (n, D), x (n, C) or (1, C), times (n,) or (1,) -> velocity (n, D), sum_f d v_f / d theta_f (n,)."""
    dev = _lib.require_device(theta_t, x, times, net.flat_params)
    n = theta_t.shape[0]
    v = torch.empty_like(theta_t) if v_out is None else v_out
    div = torch.empty(n, dtype=torch.float32, device=dev) if div_out is None else div_out
    if n == 0:
        return v, div
    with torch.cuda.device(dev):
        rc = _lib.load().sbi_amd_fmpe_velocity_div(
            net.hyper.c_config(), _lib.ptr(packed_weights(net)), _lib.ptr(net.zstats), _lib.ptr(theta_t),
            _lib.ptr(x), x.shape[0], _lib.ptr(times), times.numel(), n, _lib.ptr(v), _lib.ptr(div),
            _lib.current_stream(dev))
    _lib.check(rc, "fmpe_velocity_div")
    return v, div


def cfm_loss(net: VectorFieldMLPParams, theta: Tensor, x: Tensor, times: Tensor, noise: Tensor) -> Tensor:
    dev = _lib.require_device(theta, x, times, noise, net.flat_params)
    n = theta.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=dev)
    if n == 0:
        return out
    with torch.cuda.device(dev):
        rc = _lib.load().sbi_amd_fmpe_loss(
            net.hyper.c_config(), _lib.ptr(packed_weights(net)), _lib.ptr(net.zstats), _lib.ptr(theta), _lib.ptr(x),
            x.shape[0], _lib.ptr(times), _lib.ptr(noise), n, _lib.ptr(out), _lib.current_stream(dev))
    _lib.check(rc, "fmpe_loss")
    return out


def train_workspace(net: VectorFieldMLPParams, n: int, device, workspace: Optional[Tensor] = None) -> Tensor:
    need = _lib.load().sbi_amd_fmpe_train_workspace_floats(net.hyper.c_config(), n)
    if need < 0:
        _lib.check(int(need), "fmpe_train_workspace_floats")
    if workspace is not None and workspace.numel() >= need and workspace.device == torch.device(device):
        return workspace
    return torch.empty(int(need), dtype=torch.float32, device=device)


def loss_fwd_bwd(net: VectorFieldMLPParams, theta: Tensor, x: Tensor, times: Tensor, noise: Tensor,
                 row_weight: Optional[Tensor], uniform_weight: float, grad_out: Tensor,
                 workspace: Optional[Tensor] = None) -> Tensor:
    """Per-row CFM losses; ``grad_out`` <- d/dparams sum_i w_i loss_i."""
    dev = _lib.require_device(theta, x, times, noise, net.flat_params, grad_out, row_weight)
    n = theta.shape[0]
    ws = train_workspace(net, n, dev, workspace)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().sbi_amd_fmpe_loss_fwd_bwd(
            net.hyper.c_config(), _lib.ptr(net.flat_params), _lib.ptr(packed_weights(net)), _lib.ptr(net.zstats),
            _lib.ptr(theta), _lib.ptr(x), x.shape[0], _lib.ptr(times), _lib.ptr(noise), n, _lib.ptr(row_weight),
            float(uniform_weight), _lib.ptr(out), _lib.ptr(grad_out), _lib.ptr(ws), _lib.current_stream(dev))
    _lib.check(rc, "fmpe_loss_fwd_bwd")
    return out


class _CFMLossFn(torch.autograd.Function):
    """Autograd bridge: per-row loss whose backward hands sum_i g_i dloss_i/dparams to ``flat_params.grad``
    (the fused kernels run again with the incoming g as row weights)."""

    @staticmethod
    def forward(ctx, flat_params, net, theta, x, times, noise):
        ctx.net, ctx.args = net, (theta, x, times, noise)
        return cfm_loss(net, theta, x, times, noise)

    @staticmethod
    def backward(ctx, g):
        theta, x, times, noise = ctx.args
        grad = torch.empty_like(ctx.net.flat_params.data)
        loss_fwd_bwd(ctx.net, theta, x, times, noise, g.contiguous().float(), 0.0, grad)
        return grad, None, None, None, None, None


class FlowMatchingEstimator(ConditionalEstimator):
    """Rectified-flow-matching estimator: t = 0 is data, t = 1 is N(0, I) noise
    (flowmatching_estimator.py:23-27)."""

    SCORE_DEFINED = True
    SDE_DEFINED = True
    MARGINALS_DEFINED = True

    def __init__(self, net: VectorFieldMLPParams, input_shape: torch.Size, condition_shape: torch.Size,
                 t_min: float = 0.0, t_max: float = 1.0):
        super().__init__(input_shape, condition_shape)
        if len(input_shape) != 1 or len(condition_shape) != 1:
            raise NotImplementedError("sbi_amd FMPE: theta and x must be flat vectors (1-D event shapes)")
        self.net = net
        self.t_min, self.t_max = t_min, t_max
        self.noise_scale = net.hyper.noise_scale
        self.register_buffer("mean_base", torch.zeros(1, *input_shape))
        self.register_buffer("std_base", torch.ones(1, *input_shape))

    @property
    def embedding_net(self) -> Optional[nn.Module]:
        return None

    @property
    def mean_0(self) -> Tensor:
        return self.net.zstats[: self.net.hyper.D]

    @property
    def std_0(self) -> Tensor:
        return self.net.zstats[self.net.hyper.D : 2 * self.net.hyper.D]

    def solve_schedule(self, steps: int, t_min: Optional[float] = None, t_max: Optional[float] = None) -> Tensor:
        """Linear time grid from t_max down to t_min (estimators/base.py solve_schedule)."""
        t_min = self.t_min if t_min is None else t_min
        t_max = self.t_max if t_max is None else t_max
        return torch.linspace(t_max, t_min, steps, device=self.net.zstats.device)

    # ------------------------------------------------------------------ forward / loss
    def forward(self, input: Tensor, condition: Tensor, time: Tensor) -> Tensor:
        """Velocity in original space; input ``(*batch, D)``, condition ``(*batch_c, C)``, time broadcastable to
        the batch shape (flowmatching_estimator.py:206-274)."""
        self._check_input_shape(input)
        self._check_condition_shape(condition)
        bshape = torch.broadcast_shapes(input.shape[:-1], condition.shape[:-1])
        D, C = self.input_shape[0], self.condition_shape[0]
        th = input.to(torch.float32).expand(*bshape, D).reshape(-1, D).contiguous()
        cond = condition.to(torch.float32)
        if cond.numel() == C:
            c2 = cond.reshape(1, C).contiguous()
        else:
            c2 = cond.expand(*bshape, C).reshape(-1, C).contiguous()
        t = torch.as_tensor(time, dtype=torch.float32, device=th.device)
        t2 = t.reshape(1).contiguous() if t.numel() == 1 else t.expand(bshape).reshape(-1).contiguous()
        v = velocity(self.net, th, c2, t2)
        return v.reshape(*bshape, D)

    def ode_fn(self, input: Tensor, condition: Tensor, times: Tensor) -> Tensor:
        return self.forward(input, condition, times)

    def ode_fn_and_divergence(self, input: Tensor, condition: Tensor, times: Tensor, v_out: Optional[Tensor] = None,
                              div_out: Optional[Tensor] = None):
        """``ode_fn`` and the exact trace of its Jacobian with respect to ``input`` -- the augmented right-hand side
        zuko's ``FreeFormJacobianTransform(exact=True)`` integrates for ``VectorFieldPosterior.log_prob``
        (samplers/ode_solvers/zuko_ode.py:100-124).  input (n, D), condition (n, C) or (1, C), times 1 or n entries."""
        D, C = self.input_shape[0], self.condition_shape[0]
        th = input.to(torch.float32).reshape(-1, D).contiguous()
        cond = condition.to(torch.float32).reshape(-1, C).contiguous()
        if cond.shape[0] not in (1, th.shape[0]):
            raise ValueError(f"condition batch {cond.shape[0]} does not match input batch {th.shape[0]}")
        t = torch.as_tensor(times, dtype=torch.float32, device=th.device).reshape(-1).contiguous()
        if t.numel() not in (1, th.shape[0]):
            raise ValueError(f"times has {t.numel()} entries for {th.shape[0]} rows")
        return velocity_and_divergence(self.net, th, cond, t, v_out, div_out)

    def score(self, input: Tensor, condition: Tensor, t: Tensor) -> Tensor:
        """flowmatching_estimator.py:374-399."""
        v = self(input, condition, t)
        return (-(1 - t) * v - input) / (t + self.noise_scale)

    def loss(self, input: Tensor, condition: Tensor, times: Optional[Tensor] = None, noise: Optional[Tensor] = None,
             **kwargs) -> Tensor:
        """Per-row conditional-flow-matching loss (flowmatching_estimator.py:276-347).  ``times`` ~ U[0, 1] and
        ``noise`` ~ N(0, I) are drawn here when not given (in sbi's order: times, then noise).  Differentiable
        with respect to the parameters through the fused backward kernels."""
        self._check_input_shape(input)
        self._check_condition_shape(condition)
        if input.dim() != 2 or condition.dim() != 2 or input.shape[0] != condition.shape[0]:
            raise ValueError("sbi_amd FMPE loss expects input (B, D) and condition (B, C)")
        theta = input.to(torch.float32).contiguous()
        x = condition.to(torch.float32).contiguous()
        if times is None:
            times = torch.rand(theta.shape[0], device=theta.device, dtype=torch.float32)
        if noise is None:
            noise = torch.randn_like(theta)
        times = times.to(torch.float32).reshape(-1).contiguous()
        noise = noise.to(torch.float32).contiguous()
        if torch.is_grad_enabled() and self.net.flat_params.requires_grad:
            return _CFMLossFn.apply(self.net.flat_params, self.net, theta, x, times, noise)
        return cfm_loss(self.net, theta, x, times, noise)


def build_flow_matching_estimator(batch_theta: Tensor, batch_x: Tensor, z_score_theta: Optional[str] = "independent",
                                  z_score_x: Optional[str] = "independent", hidden_features: int = 100,
                                  num_layers: int = 5, time_embedding_dim: int = 32,
                                  sinusoidal_max_freq: float = 1000.0, noise_scale: float = 1e-3,
                                  **unsupported) -> FlowMatchingEstimator:
    """``build_vector_field_estimator(..., estimator_type="flow", net="mlp")``
    (vector_field_nets.py:136-339) for the configuration family the kernels implement."""
    from sbi_amd.utils.sbiutils import standardizing_stats, z_score_parser, z_standardization

    for k, v in unsupported.items():
        defaults = dict(net="mlp", model="mlp", time_emb_type="sinusoidal", gaussian_baseline=False,
                        compose_standardization=False, layer_norm=True, skip_connections=True)
        if k == "embedding_net" and (v is None or isinstance(v, nn.Identity)):
            continue
        if k in defaults and v == defaults[k]:
            continue
        raise NotImplementedError(f"sbi_amd FMPE: option {k}={v!r} is outside the HIP path (default MLP only)")
    if batch_theta.dim() != 2 or batch_x.dim() != 2:
        raise NotImplementedError("sbi_amd FMPE: theta and x must be (N, D) / (N, C)")
    D, C = batch_theta.shape[1], batch_x.shape[1]
    zt, structured_t = z_score_parser(z_score_theta)
    zx, structured_x = z_score_parser(z_score_x)
    if zt:
        mean_0, std_0 = z_standardization(batch_theta, structured_t)
    else:
        mean_0, std_0 = torch.zeros(D), torch.ones(D)
    if zx:
        x_mean, x_std = standardizing_stats(batch_x, structured_x)
    else:
        x_mean, x_std = torch.zeros(C), torch.ones(C)
    zstats = torch.cat([mean_0.reshape(-1).expand(D).cpu().float(), std_0.reshape(-1).expand(D).cpu().float(),
                        x_mean.reshape(-1).expand(C).cpu().float(), x_std.reshape(-1).expand(C).cpu().float()])
    hyper = FMPEHyper(D=D, C=C, hidden_features=hidden_features, num_layers=num_layers,
                      time_embedding_dim=time_embedding_dim, sinusoidal_max_freq=sinusoidal_max_freq,
                      noise_scale=noise_scale)
    net = VectorFieldMLPParams(hyper, zstats)
    return FlowMatchingEstimator(net, torch.Size([D]), torch.Size([C]))
