"""Posterior that samples a flow-matching estimator by solving its probability-flow ODE on the device.

Mirror of sbi's ``VectorFieldPosterior`` for ``sample_with="ode"``
(sbi/inference/posteriors/vector_field_posterior.py:155-330, 436-466): draw theta_1 ~ N(mean_base, std_base),
integrate d theta / dt = v(theta, t; x_o) from t_max to t_min, reject draws outside the prior support.
``log_prob`` integrates the same ODE in the other direction together with the exact divergence of the vector field
(zuko's exact-trace transform in sbi; here one HIP launch returns velocity and Jacobian trace).
"""

from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from sbi_amd.samplers.ode_solvers import odeint_dopri5
from sbi_amd.utils.sbiutils import within_support


class VectorFieldPosterior:
    def __init__(self, vector_field_estimator, prior, device: Optional[str] = None, atol: float = 1e-6,
                 rtol: float = 1e-5, max_sampling_batch_size: int = 100_000):
        self.vector_field_estimator = vector_field_estimator
        self.prior = prior
        self._device = device or str(next(vector_field_estimator.parameters()).device)
        self.atol, self.rtol = atol, rtol
        self.max_sampling_batch_size = max_sampling_batch_size
        self._x: Optional[Tensor] = None

    @property
    def default_x(self) -> Optional[Tensor]:
        return self._x

    def set_default_x(self, x: Tensor) -> "VectorFieldPosterior":
        self._x = self._process_x(x)
        return self

    def _process_x(self, x: Tensor) -> Tensor:
        x = torch.as_tensor(x, dtype=torch.float32)
        cshape = self.vector_field_estimator.condition_shape
        if x.dim() == len(cshape):
            x = x.unsqueeze(0)
        if x.shape[0] != 1 or x.shape[1:] != cshape:
            raise ValueError(f"expected one observation of shape {tuple(cshape)}, got {tuple(x.shape)}; use "
                             "sample_batched for several observations")
        return x.to(self._device)

    def _x_else_default_x(self, x: Optional[Tensor]) -> Tensor:
        if x is not None:
            return self._process_x(x)
        if self._x is None:
            raise ValueError("Context `x` needed when a default has not been set. Use `.set_default_x(x)` or pass "
                             "`x` explicitly.")
        return self._x

    @torch.no_grad()
    def sample_via_ode(self, num_samples: int, x: Tensor) -> Tensor:
        est = self.vector_field_estimator
        D = est.input_shape[0]
        eps = est.mean_base + est.std_base * torch.randn(num_samples, D, device=self._device)
        cond = x if x.shape[0] == num_samples else x[:1]
        return odeint_dopri5(lambda t, y: est.ode_fn(y, cond, t), eps.contiguous(), est.t_max, est.t_min,
                             atol=self.atol, rtol=self.rtol)

    @torch.no_grad()
    def sample(self, sample_shape=torch.Size(), x: Optional[Tensor] = None, max_sampling_batch_size: Optional[int] = None,
               sample_with: Optional[str] = None, show_progress_bars: bool = False,
               reject_outside_prior: bool = True, **unsupported) -> Tensor:
        if sample_with not in (None, "ode"):
            raise NotImplementedError("sbi_amd FMPE posterior samples with the probability-flow ODE only")
        x = self._x_else_default_x(x)
        num = int(torch.Size(sample_shape).numel())
        cap = max_sampling_batch_size or self.max_sampling_batch_size
        out, have, tries = [], 0, 0
        while have < num:
            n = min(cap, max(num - have, 16))
            draws = self.sample_via_ode(n, x)
            if reject_outside_prior and self.prior is not None:
                draws = draws[within_support(self.prior, draws)]
            out.append(draws)
            have += draws.shape[0]
            tries += 1
            if tries > 1000:
                raise RuntimeError("VectorFieldPosterior.sample: acceptance rate too low")
        return torch.cat(out)[:num].reshape(*torch.Size(sample_shape), -1)

    @torch.no_grad()
    def sample_batched(self, sample_shape, x: Tensor, **kwargs) -> Tensor:
        """(sample_shape, batch, D): every observation integrates its own draws in one batched ODE solve."""
        x = torch.as_tensor(x, dtype=torch.float32).to(self._device)
        num = int(torch.Size(sample_shape).numel())
        B = x.shape[0]
        xs = x.repeat_interleave(num, dim=0).contiguous()
        draws = self.sample_via_ode(num * B, xs)
        return draws.reshape(B, num, -1).permute(1, 0, 2).reshape(*torch.Size(sample_shape), B, -1)

    @torch.no_grad()
    def log_prob(self, theta: Tensor, x: Optional[Tensor] = None, track_gradients: bool = False,
                 ode_kwargs: Optional[dict] = None, max_batch_size: Optional[int] = None) -> Tensor:
        r"""``(len(theta),)`` log posterior density :math:`\log p(\theta | x)` through the probability-flow ODE, -inf
        outside the prior support (sbi/inference/posteriors/vector_field_posterior.py:467-504 ->
        potentials/vector_field_potential.py:149-207): integrate the augmented state
        :math:`(\theta_t, \ell_t)' = (v, \nabla \cdot v)` from ``t_min`` (data) to ``t_max`` (noise) and add the
        base log-density of the end point -- zuko's ``FreeFormJacobianTransform(exact=True)`` inside a
        ``NormalizingFlow`` with ``DiagNormal(mean_base, std_base)`` (samplers/ode_solvers/zuko_ode.py:100-124).  The
        exact Jacobian trace comes out of the same HIP launch as the velocity (``sbi_amd_fmpe_velocity_div``); the
        state of all rows, log-det included, stays on the device for the whole solve (``odeint_dopri5``).

        ``x`` with more than one row is a set of iid observations (vector_field_posterior.py:489-497,
        vector_field_potential.py:175-192): the log-densities of the per-observation flows are summed and
        ``(n_iid - 1) * prior.log_prob(theta)`` is subtracted; that needs a prior.

        ``ode_kwargs``: ``atol`` / ``rtol`` (defaults: the posterior's, sbi's 1e-6 / 1e-5); ``exact=False`` (zuko's
        Hutchinson estimate) is not offered."""
        if track_gradients:
            raise NotImplementedError("sbi_amd: log_prob of the flow-matching posterior does not track gradients")
        kw = dict(ode_kwargs or {})
        if not kw.pop("exact", True):
            raise NotImplementedError("sbi_amd: only the exact Jacobian trace (zuko's exact=True, sbi's default)")
        atol, rtol = float(kw.pop("atol", self.atol)), float(kw.pop("rtol", self.rtol))
        if kw:
            raise TypeError(f"unsupported ode_kwargs: {sorted(kw)}")
        est = self.vector_field_estimator
        if x is not None and torch.as_tensor(x).dim() == len(est.condition_shape) + 1 and torch.as_tensor(x).shape[0] > 1:
            xs = torch.as_tensor(x, dtype=torch.float32).to(self._device)
            if xs.shape[1:] != est.condition_shape:
                raise ValueError(f"expected observations of shape {tuple(est.condition_shape)}, got {tuple(xs.shape[1:])}")
            if self.prior is None:
                raise AssertionError("Prior is required for evaluating log_prob with iid observations.")
        else:
            xs = self._x_else_default_x(x)
        D = est.input_shape[0]
        theta = torch.as_tensor(theta, dtype=torch.float32)
        if theta.dim() == 1:
            theta = theta.unsqueeze(0)
        if theta.dim() != 2 or theta.shape[1] != D:
            raise ValueError(f"theta must have shape (batch, {D}), got {tuple(theta.shape)}")
        theta = theta.to(self._device).contiguous()
        if theta.shape[0] == 0:
            return torch.empty(0, dtype=torch.float32, device=self._device)
        cap = max_batch_size or self.max_sampling_batch_size
        log_probs = torch.zeros(theta.shape[0], dtype=torch.float32, device=self._device)
        for i_obs in range(xs.shape[0]):          # one flow per iid observation (one observation: the usual case)
            x_i = xs[i_obs : i_obs + 1]
            log_probs = log_probs + torch.cat([self._log_prob_via_ode(theta[i : i + cap], x_i, atol, rtol)
                                               for i in range(0, theta.shape[0], cap)])
        if xs.shape[0] > 1:
            log_probs = log_probs - (xs.shape[0] - 1) * self.prior.log_prob(theta).reshape(-1)
        if self.prior is not None:
            inside = within_support(self.prior, theta)
            log_probs = torch.where(inside, log_probs, torch.full_like(log_probs, float("-inf")))
        return log_probs

    def _log_prob_via_ode(self, theta: Tensor, x: Tensor, atol: float, rtol: float) -> Tensor:
        est = self.vector_field_estimator
        n, D = theta.shape
        if n == 0:
            return torch.empty(0, device=theta.device)
        # flat augmented state [theta (n x D) | ladj (n)]: both halves are contiguous views, so the right-hand side
        # kernel reads theta_t from, and writes velocity and divergence straight into, the solver's buffers
        y0 = torch.cat([theta.reshape(-1), torch.zeros(n, dtype=torch.float32, device=theta.device)])

        def rhs(t: Tensor, y: Tensor) -> Tensor:
            out = torch.empty_like(y)
            est.ode_fn_and_divergence(y[: n * D].view(n, D), x, t, v_out=out[: n * D].view(n, D), div_out=out[n * D :])
            return out

        y1 = odeint_dopri5(rhs, y0, est.t_min, est.t_max, atol=atol, rtol=rtol)
        z = (y1[: n * D].view(n, D) - est.mean_base) / est.std_base
        base = (-0.5 * z * z - torch.log(est.std_base) - 0.9189385332046727).sum(-1)
        return base + y1[n * D :]
