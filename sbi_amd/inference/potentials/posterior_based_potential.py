"""Potential = estimator log-prob masked by the prior support.

Mirror of sbi/inference/potentials/posterior_based_potential.py:26-191 (and the
``BasePotential`` x_o handling, base_potential.py:16-105) for the NPE case: a
single ``x_o`` is broadcast against all thetas WITHOUT materialising copies
(the kernel reads ``x[n % x_rows]``); a batch of x_o is evaluated pairwise.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.distributions import Distribution

from sbi_amd.neural_nets.estimators.base import ConditionalDensityEstimator
from sbi_amd.neural_nets.estimators.shape_handling import reshape_to_batch_event, reshape_to_sample_batch_event
from sbi_amd.utils.sbiutils import within_support
from sbi_amd.utils.torchutils import ensure_theta_batched


class PosteriorBasedPotential:
    def __init__(self, posterior_estimator: ConditionalDensityEstimator, prior: Distribution,
                 x_o: Optional[Tensor] = None, device: str = "cpu"):
        self.posterior_estimator = posterior_estimator
        self.prior = prior
        self.device = device
        self._x_o: Optional[Tensor] = None
        self.x_is_iid = False
        self.posterior_estimator.eval()
        if x_o is not None:
            self.set_x(x_o)

    # -- x_o handling -------------------------------------------------------------------
    def set_x(self, x_o: Optional[Tensor], x_is_iid: Optional[bool] = False) -> None:
        if x_is_iid and x_o is not None and x_o.dim() > 1 and x_o.shape[0] > 1:
            raise NotImplementedError(
                "For NPE, iid observations need a permutation-invariant embedding net, which is outside "
                "this path (posterior_based_potential.py:74-83)."
            )
        self._x_o = None if x_o is None else x_o.to(self.device)

    @property
    def x_o(self) -> Tensor:
        if self._x_o is None:
            raise ValueError("No observed data is available.")
        return self._x_o

    @x_o.setter
    def x_o(self, x_o: Optional[Tensor]) -> None:
        self.set_x(x_o)

    def return_x_o(self) -> Optional[Tensor]:
        return self._x_o

    def to(self, device: str) -> "PosteriorBasedPotential":
        self.device = device
        self.posterior_estimator.to(device)
        if hasattr(self.prior, "to"):
            self.prior = self.prior.to(device)
        if self._x_o is not None:
            self._x_o = self._x_o.to(device)
        return self

    # -- evaluation ---------------------------------------------------------------------
    def __call__(self, theta: Tensor, track_gradients: bool = True) -> Tensor:
        """log p(theta | x_o) where theta is in the prior support, -inf elsewhere; shape (num_thetas,)."""
        theta = ensure_theta_batched(torch.as_tensor(theta)).to(self.device)
        est = self.posterior_estimator
        x = reshape_to_batch_event(self.x_o, event_shape=est.condition_shape)
        with torch.set_grad_enabled(track_gradients):
            if x.shape[0] == 1:
                # (N,1,D) thetas against one (1,C) condition (posterior_based_potential.py:170-178)
                theta_sbe = reshape_to_sample_batch_event(theta, event_shape=theta.shape[1:], leading_is_sample=True)
                lp = est.log_prob(theta_sbe, condition=x).squeeze(1)
            else:
                if theta.shape[0] != x.shape[0]:
                    raise ValueError(
                        f"Batch shape of theta {theta.shape[0]} and x_o {x.shape[0]} must match for batched "
                        "evaluation."
                    )
                lp = est.log_prob(theta.unsqueeze(0), condition=x).squeeze(0)
            in_support = within_support(self.prior, theta)
            lp = torch.where(in_support, lp, torch.tensor(float("-inf"), dtype=torch.float32, device=lp.device))
        return lp


def posterior_estimator_based_potential(posterior_estimator, prior, x_o: Optional[Tensor],
                                        enable_transform: bool = True) -> Tuple[PosteriorBasedPotential, object]:
    """(potential_fn, theta_transform).  The transform is the identity here: NPE samples
    directly from the estimator; unconstrained-space transforms serve MCMC/VI/MAP callers."""
    device = str(next(posterior_estimator.parameters()).device)
    potential_fn = PosteriorBasedPotential(posterior_estimator, prior, x_o, device=device)
    return potential_fn, torch.distributions.transforms.identity_transform
