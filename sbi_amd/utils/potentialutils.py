"""sbi/utils/potentialutils.py:15-51."""

from __future__ import annotations

from typing import Callable

import torch
from torch import Tensor

from sbi_amd.utils.torchutils import ensure_theta_batched


def transformed_potential(theta: Tensor, potential_fn: Callable, theta_transform, device: str,
                          track_gradients: bool = False) -> Tensor:
    """Potential of parameters given in TRANSFORMED (unconstrained) space: potential(T^-1(u)) - log|det dT|."""
    transformed_theta = ensure_theta_batched(torch.as_tensor(theta, dtype=torch.float32)).to(device)
    theta = theta_transform.inv(transformed_theta)
    log_abs_det = theta_transform.log_abs_det_jacobian(theta, transformed_theta)
    posterior_potential = potential_fn(theta, track_gradients=track_gradients)
    return posterior_potential.to(device) - log_abs_det.to(device)
