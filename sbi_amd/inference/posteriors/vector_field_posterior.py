"""Posterior that samples a flow-matching estimator by solving its probability-flow ODE on the device.

Mirror of sbi's ``VectorFieldPosterior`` for ``sample_with="ode"``
(sbi/inference/posteriors/vector_field_posterior.py:155-330, 436-466): draw theta_1 ~ N(mean_base, std_base),
integrate d theta / dt = v(theta, t; x_o) from t_max to t_min, reject draws outside the prior support.
``log_prob`` needs the divergence of the vector field along the trajectory (zuko's exact-trace transform in
sbi); that is not part of this path and raises.
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

    def log_prob(self, theta: Tensor, x: Optional[Tensor] = None, **kwargs) -> Tensor:
        raise NotImplementedError("log_prob of the flow-matching posterior (trace of the Jacobian along the ODE) "
                                  "is outside the HIP path")
