"""Estimator contracts -- the drop-in boundary of the hot path.

Mirrors sbi/neural_nets/estimators/base.py:17-306 (same method names, argument
meaning, shapes and error behaviour) so that anything written against sbi's
``ConditionalDensityEstimator`` runs against ours.
"""

from abc import ABC, abstractmethod
from typing import Optional, Protocol, Tuple

import torch
from torch import Tensor, nn


class ConditionalEstimatorBuildFn(Protocol):
    """``build_fn(theta, x) -> ConditionalEstimator`` (base.py:17-32)."""

    def __call__(self, theta: Tensor, x: Tensor) -> "ConditionalEstimator": ...


class ConditionalEstimator(nn.Module, ABC):
    """An `nn.Module` that knows the event shapes of its input and of what it is conditioned on."""

    def __init__(self, input_shape: Tuple, condition_shape: Tuple) -> None:
        super().__init__()
        self._event_shapes = (torch.Size(input_shape), torch.Size(condition_shape))

    input_shape = property(lambda self: self._event_shapes[0], doc="event shape of `input` (theta)")
    condition_shape = property(lambda self: self._event_shapes[1], doc="event shape of `condition` (x)")

    @abstractmethod
    def loss(self, input: Tensor, condition: Tensor, **kwargs) -> Tensor: ...

    @staticmethod
    def _check_event(t: Tensor, expected: torch.Size, what: str, attr: str) -> None:
        if t.dim() < len(expected):
            raise ValueError(
                f"Dimensionality of {what} is too small and does not match the expected "
                f"dimensionality {len(expected)}. It should be compatible with {attr} {expected}."
            )
        got = t.shape[t.dim() - len(expected) :] if len(expected) else torch.Size()
        if got != expected:
            raise ValueError(
                f"Shape of {what} {got} does not match the expected input dimensionality "
                f"{expected}, as provided by {attr}. Please reshape it accordingly."
            )

    def _check_condition_shape(self, condition: Tensor) -> None:
        self._check_event(condition, self.condition_shape, "condition", "condition_shape")

    def _check_input_shape(self, input: Tensor) -> None:
        self._check_event(input, self.input_shape, "input", "input_shape")

    def _broadcast_dims(self, input: Tensor, condition: Tensor) -> Tuple[Tensor, int, int, bool]:
        """Shape bookkeeping of base.py:142-198 WITHOUT materialising the expands.

        Returns ``(input with sample dim, sample_dim, batch_dim, condition_has_sample_dim)``.
        """
        if input.dim() <= len(self.input_shape) + 1:
            input = input.unsqueeze(0)
        sample_dim, input_batch = input.shape[0], input.shape[1]
        cond_has_sample = condition.dim() > len(self.condition_shape) + 1
        cond_batch = condition.shape[1] if cond_has_sample else condition.shape[0]
        try:
            batch_dim = torch.broadcast_shapes((input_batch,), (cond_batch,))[0]
        except RuntimeError as err:
            raise RuntimeError(
                "Expected `input` and `condition` to have broadcastable batch "
                "dimensions: their batch sizes must match, or one of them must be 1. "
                f"Got input={input_batch} and condition={cond_batch}."
            ) from err
        return input, sample_dim, batch_dim, cond_has_sample

    def _broadcast_and_align(self, input: Tensor, condition: Tensor) -> Tuple[Tensor, Tensor, int]:
        """Both arguments as (sample, batch, *event) views of one common (sample, batch) grid, plus the batch size
        (base.py:142-198; expanded VIEWS -- whoever reshapes them pays for the copy, the kernels never do)."""
        input, n_sample, n_batch, cond_has_sample = self._broadcast_dims(input, condition)
        grid = (n_sample, n_batch)
        if not cond_has_sample:
            condition = condition.unsqueeze(0)                   # (1, 1 | batch, *event)
        return (torch.broadcast_to(input, grid + tuple(self.input_shape)),
                torch.broadcast_to(condition, grid + tuple(self.condition_shape)), n_batch)


class ConditionalDensityEstimator(ConditionalEstimator):
    """``log_prob -> (S,B)``, ``loss -> (B,)``, ``sample -> (*shape, B, *event)``."""

    def __init__(self, net: nn.Module, input_shape: torch.Size, condition_shape: torch.Size) -> None:
        super().__init__(input_shape, condition_shape)
        self.net = net

    @property
    def embedding_net(self) -> Optional[nn.Module]:
        return None

    @abstractmethod
    def log_prob(self, input: Tensor, condition: Tensor, **kwargs) -> Tensor: ...

    @abstractmethod
    def loss(self, input: Tensor, condition: Tensor, **kwargs) -> Tensor: ...

    @abstractmethod
    def sample(self, sample_shape: torch.Size, condition: Tensor, **kwargs) -> Tensor: ...

    def sample_and_log_prob(self, sample_shape: torch.Size, condition: Tensor, **kwargs) -> Tuple[Tensor, Tensor]:
        """Default: draw, then evaluate the draws (flows override it with the one-pass form)."""
        drawn = self.sample(sample_shape, condition, **kwargs)
        return drawn, self.log_prob(drawn, condition, **kwargs)
