"""Linear-Gaussian toy simulators and analytic posteriors (test / benchmark inputs).

Same functions, arguments and maths as sbi/simulators/linear_gaussian.py:15-105:
``x = shift + theta + chol(cov) eps``; the product of the Gaussian likelihood and
a Gaussian prior gives the reference posterior the C2ST checks compare against.
"""

from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor
from torch.distributions import MultivariateNormal

from sbi_amd.utils.torchutils import atleast_2d


def diagonal_linear_gaussian(theta: Tensor, std: float = 1.0) -> Tensor:
    """Gaussian likelihood with diagonal covariance: ``theta + std * eps``."""
    return theta + std * torch.randn_like(theta)


def linear_gaussian(theta: Tensor, likelihood_shift: Tensor, likelihood_cov: Tensor,
                    num_discarded_dims: int = 0) -> Tensor:
    """``x ~ N(likelihood_shift + theta, likelihood_cov)``, optionally on the leading dims only."""
    theta = torch.as_tensor(theta)
    if num_discarded_dims:
        theta = theta[:, :-num_discarded_dims]
    chol = torch.linalg.cholesky(likelihood_cov)
    return likelihood_shift + theta + torch.mm(chol, torch.randn_like(theta).T).T


def multiply_gaussian_pdfs(mu1: Tensor, s1: Tensor, mu2: Tensor, s2: Tensor) -> Tuple[Tensor, Tensor]:
    """Mean and covariance of the (unnormalised) product N(mu1,s1) * N(mu2,s2)."""
    inv_s1s2 = torch.inverse(s1 + s2)
    product_mean = torch.mv(torch.mm(s2, inv_s1s2), mu1) + torch.mv(torch.mm(s1, inv_s1s2), mu2)
    product_cov = torch.mm(torch.mm(s1, inv_s1s2), s2)
    return product_mean, product_cov


def true_posterior_linear_gaussian_mvn_prior(x_o: Tensor, likelihood_shift: Tensor, likelihood_cov: Tensor,
                                             prior_mean: Tensor, prior_cov: Tensor) -> MultivariateNormal:
    """Analytic posterior for Gaussian likelihood (iid trials in x_o rows) and Gaussian prior."""
    x_o = atleast_2d(x_o)
    num_trials = x_o.shape[0]
    likelihood_mean = x_o.mean(0) - likelihood_shift
    mean, cov = multiply_gaussian_pdfs(likelihood_mean, 1 / num_trials * likelihood_cov, prior_mean, prior_cov)
    return MultivariateNormal(mean, cov)
