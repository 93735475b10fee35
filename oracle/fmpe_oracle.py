"""CPU oracle for the FMPE (flow-matching) vector-field path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(sbi_amd/) never does.

It restates, in plain fp32 PyTorch on the CPU, what sbi computes for its default flow-matching estimator:

* the vector-field MLP            sbi/neural_nets/net_builders/vector_field_nets.py:610-719  (VectorFieldMLP)
* its sinusoidal time embedding   vector_field_nets.py:367-421  (built with max_freq 1000, :1276-1343)
* time-dependent z-scoring, velocity normalisation, the conditional-flow-matching loss and the velocity
  returned to ODE solvers        sbi/neural_nets/estimators/flowmatching_estimator.py:120-164, 206-274, 276-347
* x standardisation in front      vector_field_nets.py:271-276 + sbi/utils/sbiutils.py:418-428

PARITY PINNED: unlike the nflows boundary of the NSF path, every piece above is sbi's own Python, importable in
the build container; tools/make_golden_fmpe.py runs the real classes and tests/test_golden_fmpe.py holds this
restatement to their outputs (loss, velocity, parameter gradients).

The parameters are kept under sbi's state-dict names so a reference `state_dict()` loads directly.
"""

from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import torch
from torch import Tensor, nn


def sinusoidal_frequencies(embed_dim: int, max_freq: float) -> Tensor:
    """`div_term` of SinusoidalTimeEmbedding (vector_field_nets.py:387-390)."""
    return torch.exp(torch.arange(0, embed_dim, 2) * (-math.log(max_freq) / embed_dim))


def gelu_exact(v: Tensor) -> Tensor:
    """nn.GELU() with the default `approximate='none'`: v * Phi(v)."""
    return 0.5 * v * (1.0 + torch.erf(v * (1.0 / math.sqrt(2.0))))


class FMPEOracle(nn.Module):
    """Flow-matching estimator with the default MLP, one flat event dimension for theta and (embedded) x."""

    def __init__(self, D: int, C: int, H: int = 100, L: int = 5, E: int = 32, max_freq: float = 1000.0,
                 noise_scale: float = 1e-3, ln_eps: float = 1e-5):
        super().__init__()
        self.D, self.C, self.H, self.L, self.E = D, C, H, L, E
        self.noise_scale, self.ln_eps = noise_scale, ln_eps
        p = OrderedDict()

        def lin(name, out_f, in_f):
            p[f"net.{name}.weight"] = nn.Parameter(torch.empty(out_f, in_f))
            p[f"net.{name}.bias"] = nn.Parameter(torch.empty(out_f))

        lin("input_layer", H, D)
        lin("condition_layer", H, C)
        lin("input_merge_layer", H, 2 * H)
        for i in range(L):
            lin(f"layers.{i}", H, H)
        for i in range(L):
            p[f"net.layers_norm.{i}.weight"] = nn.Parameter(torch.ones(H))
            p[f"net.layers_norm.{i}.bias"] = nn.Parameter(torch.zeros(H))
        lin("time_linear_layer", H, E)
        lin("output_layer", D, H)
        self.p = nn.ParameterDict({k.replace(".", "/"): v for k, v in p.items()})
        self.register_buffer("div_term", sinusoidal_frequencies(E, max_freq))
        self.register_buffer("mean_0", torch.zeros(D))
        self.register_buffer("std_0", torch.ones(D))
        self.register_buffer("x_mean", torch.zeros(C))
        self.register_buffer("x_std", torch.ones(C))
        for k, v in self.p.items():   # placeholder init; real values come from load_reference_state_dict
            nn.init.normal_(v, std=0.1)

    # ------------------------------------------------------------------ state
    def W(self, name: str) -> Tensor:
        return self.p[("net." + name).replace(".", "/")]

    def load_reference_state_dict(self, sd: Dict[str, Tensor]) -> None:
        """Accepts `FlowMatchingEstimator.state_dict()` of the reference (or of sbi_amd's estimator)."""
        with torch.no_grad():
            for k, v in self.p.items():
                v.copy_(sd[k.replace("/", ".")])
            self.mean_0.copy_(sd["mean_0"])
            self.std_0.copy_(sd["std_0"])
            if "_embedding_net.0._mean" in sd:
                self.x_mean.copy_(sd["_embedding_net.0._mean"])
                self.x_std.copy_(sd["_embedding_net.0._std"])
            if "net.time_emb.div_term" in sd:
                self.div_term.copy_(sd["net.time_emb.div_term"])

    # ------------------------------------------------------------------ pieces
    def time_features(self, t: Tensor) -> Tensor:
        """(N,) -> (N, E): even columns sin(t w_k), odd columns cos(t w_k)  (vector_field_nets.py:414-419)."""
        ang = t[:, None] * self.div_term[None, :]
        return torch.stack([torch.sin(ang), torch.cos(ang)], dim=-1).reshape(t.shape[0], -1)

    def marginal_stats(self, t: Tensor):
        """mu_t, std_t of theta_t (flowmatching_estimator.py:140-147)."""
        a = (1.0 - t)[:, None]
        mu = a * self.mean_0[None, :]
        var = (a * self.std_0[None, :]) ** 2 + t[:, None] ** 2 + 1e-6
        return mu, torch.sqrt(var)

    def velocity_stats(self):
        """flowmatching_estimator.py:160-164."""
        return -self.mean_0[None, :], torch.sqrt(1.0 + self.std_0[None, :] ** 2)

    def net(self, inp: Tensor, cond: Tensor, t: Tensor, trace: Optional[dict] = None) -> Tensor:
        """VectorFieldMLP.forward (vector_field_nets.py:683-719) on flat (N, .) inputs."""
        F = nn.functional
        ie = F.linear(inp, self.W("input_layer.weight"), self.W("input_layer.bias"))
        ce = F.linear(cond, self.W("condition_layer.weight"), self.W("condition_layer.bias"))
        h0 = F.linear(gelu_exact(torch.cat([ie, ce], dim=-1)), self.W("input_merge_layer.weight"),
                      self.W("input_merge_layer.bias"))
        temb = F.linear(self.time_features(t), self.W("time_linear_layer.weight"), self.W("time_linear_layer.bias"))
        h = gelu_exact(h0)
        if trace is not None:
            trace.update(ie=ie, ce=ce, h0=h0, temb=temb, u=[], h=[])
        for i in range(self.L):
            u = F.linear(h, self.W(f"layers.{i}.weight"), self.W(f"layers.{i}.bias"))
            s = gelu_exact(u) + temb + h
            mu = s.mean(-1, keepdim=True)
            var = ((s - mu) ** 2).mean(-1, keepdim=True)
            h = (s - mu) / torch.sqrt(var + self.ln_eps) * self.W(f"layers_norm.{i}.weight") \
                + self.W(f"layers_norm.{i}.bias")
            if trace is not None:
                trace["u"].append(u)
                trace["h"].append(h)
        return F.linear(h, self.W("output_layer.weight"), self.W("output_layer.bias"))

    def embed(self, x: Tensor) -> Tensor:
        return (x - self.x_mean) / self.x_std

    # ------------------------------------------------------------------ estimator interface
    def velocity(self, theta_t: Tensor, x: Tensor, t: Tensor) -> Tensor:
        """FlowMatchingEstimator.forward without the Gaussian baseline (flowmatching_estimator.py:206-274);
        theta_t (N, D), x (N, C) or (1, C), t (N,)."""
        n = theta_t.shape[0]
        c = self.embed(x).expand(n, -1)
        mu, sd = self.marginal_stats(t)
        vm, vs = self.velocity_stats()
        return self.net((theta_t - mu) / sd, c, t) * vs + vm

    def velocity_and_divergence(self, theta_t: Tensor, x: Tensor, t: Tensor):
        """`velocity` and the exact trace of its Jacobian wrt theta_t: the augmented right-hand side that
        VectorFieldPosterior.log_prob integrates (sbi/inference/posteriors/vector_field_posterior.py:467-504 ->
        potentials/vector_field_potential.py:149-207 -> samplers/ode_solvers/zuko_ode.py:100-124, where zuko's
        FreeFormJacobianTransform(exact=True) -- third-party, zuko >= 1.2, not installed here -- takes the trace with a
        batched autograd identity).  Here: one reverse pass per theta dim; rows are independent, so the gradient of
        sum_n v[n, f] picks d v[n, f] / d theta[n, :].  Pinned against the real estimator's ode_fn by
        tests/test_golden_fmpe.py (fixture key `div`)."""
        th = theta_t.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            v = self.velocity(th, x, t)
            div = torch.zeros(th.shape[0], dtype=th.dtype)
            for f in range(self.D):
                (gf,) = torch.autograd.grad(v[:, f].sum(), th, retain_graph=True)
                div = div + gf[:, f]
        return v.detach(), div.detach()

    def log_prob(self, theta: Tensor, x: Tensor, steps: int = 200, t_min: float = 0.0, t_max: float = 1.0) -> Tensor:
        """log p(theta | x) of the probability-flow ODE: integrate (theta, ladj)' = (v, div v) from t_min (data)
        to t_max (noise) and add the N(0, I) base log-density of the end point (zuko's
        NormalizingFlow.log_prob = base.log_prob(transform(theta)) + ladj; sbi builds the base as
        DiagNormal(mean_base = 0, std_base = 1), samplers/ode_solvers/zuko_ode.py:118-124).  Classical RK4 on a fixed
        grid in the module's dtype: run it on `.double()` with a few hundred steps and the result is the ODE's
        solution to ~1e-9 -- the yardstick the adaptive fp32 device solve is held to."""
        th = theta.detach().clone()
        n = th.shape[0]
        ladj = torch.zeros(n, dtype=th.dtype)
        hstep = (t_max - t_min) / steps

        def rhs(tt: float, y: Tensor):
            return self.velocity_and_divergence(y, x, torch.full((n,), tt, dtype=th.dtype))

        for i in range(steps):
            t0 = t_min + i * hstep
            k1, d1 = rhs(t0, th)
            k2, d2 = rhs(t0 + 0.5 * hstep, th + 0.5 * hstep * k1)
            k3, d3 = rhs(t0 + 0.5 * hstep, th + 0.5 * hstep * k2)
            k4, d4 = rhs(t0 + hstep, th + hstep * k3)
            th = th + hstep / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
            ladj = ladj + hstep / 6.0 * (d1 + 2 * d2 + 2 * d3 + d4)
        base = (-0.5 * th ** 2 - 0.5 * math.log(2.0 * math.pi)).sum(-1)
        return base + ladj

    def loss(self, theta: Tensor, x: Tensor, times: Tensor, noise: Tensor) -> Tensor:
        """FlowMatchingEstimator.loss (flowmatching_estimator.py:276-347) with the draws `times ~ U[0,1]` and
        `noise = theta_1 ~ N(0, I)` made explicit.  Returns the per-row loss (N,)."""
        tcol = times[:, None]
        theta_t = (1.0 - tcol) * theta + (tcol + self.noise_scale) * noise
        mu, sd = self.marginal_stats(times)
        vm, vs = self.velocity_stats()
        target = ((noise - theta) - vm) / vs
        out = self.net((theta_t - mu) / sd, self.embed(x), times)
        return ((out - target) ** 2).mean(-1)
