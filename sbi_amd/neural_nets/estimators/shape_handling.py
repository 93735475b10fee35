"""(sample, batch, event) shape conventions of the estimator contract.

Behavioural mirror of sbi/neural_nets/estimators/shape_handling.py:8-96.
"""

import torch
from torch import Tensor


def _split(t: Tensor, event_shape: torch.Size):
    n_event = len(event_shape)
    trailing = t.shape[t.dim() - n_event :] if n_event else torch.Size()
    leading = t.shape[: t.dim() - n_event]
    if trailing != torch.Size(event_shape):
        raise RuntimeError(
            "The shape of the input does not match the expected shape. Expected trailing "
            f"dimensions {tuple(event_shape)}, but got {tuple(trailing)}. This can happen "
            "when `x_o` has more (or fewer) entries than the `x` used during training."
        )
    return leading


def reshape_to_sample_batch_event(
    theta_or_x: Tensor, event_shape: torch.Size, leading_is_sample: bool = False
) -> Tensor:
    """Return a view of shape ``(sample, batch, *event)``."""
    leading = _split(theta_or_x, event_shape)
    if len(leading) == 0:
        return theta_or_x.unsqueeze(0).unsqueeze(0)
    if len(leading) == 1:
        return theta_or_x.unsqueeze(1) if leading_is_sample else theta_or_x.unsqueeze(0)
    if len(leading) == 2:
        return theta_or_x if leading_is_sample else theta_or_x.transpose(1, 0)
    raise ValueError(
        f"`len(leading_theta_or_x_shape) = {leading} > 2`. It is unclear how the "
        "additional entries should be interpreted"
    )


def reshape_to_batch_event(theta_or_x: Tensor, event_shape: torch.Size) -> Tensor:
    """Return a view of shape ``(batch, *event)``."""
    leading = _split(theta_or_x, event_shape)
    if len(leading) == 0:
        return theta_or_x.unsqueeze(0)
    if len(leading) == 1:
        return theta_or_x
    raise ValueError(
        f"`len(leading_theta_or_x_shape) = {leading} > 1`. It is unclear how the "
        "additional entries should be interpreted"
    )
