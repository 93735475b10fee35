"""Linear-Gaussian toy simulators and their analytic posterior: inputs of the C2ST / parity tests and of bench.py.

Behaviour of sbi/simulators/linear_gaussian.py:15-105 (pinned by tests/golden/reference_intree.pt):
``x = shift + theta + L eps`` with ``L L^T = cov``; under a Gaussian prior the posterior given n i.i.d. observations is
the normalised product of ``N(mean(x_o) - shift, cov / n)`` and the prior.
"""

from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor
from torch.distributions import MultivariateNormal

from sbi_amd.utils.torchutils import atleast_2d


def diagonal_linear_gaussian(theta: Tensor, std: float = 1.0) -> Tensor:
    """One observation per parameter row with independent noise of scale `std` on every coordinate."""
    noise = torch.randn_like(theta)
    return theta + std * noise


def linear_gaussian(theta: Tensor, likelihood_shift: Tensor, likelihood_cov: Tensor,
                    num_discarded_dims: int = 0) -> Tensor:
    """One observation per parameter row, ``N(theta + likelihood_shift, likelihood_cov)``; the last
    `num_discarded_dims` parameter coordinates do not enter the observation."""
    theta = torch.as_tensor(theta)
    kept = theta if num_discarded_dims == 0 else theta[:, : theta.shape[1] - num_discarded_dims]
    factor = torch.linalg.cholesky(likelihood_cov)
    correlated = (factor @ torch.randn_like(kept).T).T          # (noise drawn row-major, as the golden run drew it)
    return likelihood_shift + kept + correlated


def multiply_gaussian_pdfs(mu1: Tensor, s1: Tensor, mu2: Tensor, s2: Tensor) -> Tuple[Tensor, Tensor]:
    """(mean, covariance) of the normalised product of two Gaussian densities in the same variable:
    with ``G = (s1 + s2)^-1``: mean ``s2 G mu1 + s1 G mu2``, covariance ``s1 G s2``."""
    gain = torch.inverse(s1 + s2)
    to_first, to_second = s2 @ gain, s1 @ gain
    return to_first @ mu1 + to_second @ mu2, to_second @ s2


def true_posterior_linear_gaussian_mvn_prior(x_o: Tensor, likelihood_shift: Tensor, likelihood_cov: Tensor,
                                             prior_mean: Tensor, prior_cov: Tensor) -> MultivariateNormal:
    """The exact posterior for the rows of `x_o` read as i.i.d. observations under a Gaussian prior."""
    observations = atleast_2d(x_o)
    n = observations.shape[0]
    evidence_mean = observations.mean(dim=0) - likelihood_shift
    mean, cov = multiply_gaussian_pdfs(evidence_mean, likelihood_cov / n, prior_mean, prior_cov)
    return MultivariateNormal(mean, cov)
