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
    def __init__(self, input_shape: Tuple, condition_shape: Tuple) -> None:
        super().__init__()
        self._input_shape = torch.Size(input_shape)
        self._condition_shape = torch.Size(condition_shape)

    @property
    def input_shape(self) -> torch.Size:
        return self._input_shape

    @property
    def condition_shape(self) -> torch.Size:
        return self._condition_shape

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
        input, sample_dim, batch_dim, cond_has_sample = self._broadcast_dims(input, condition)
        input = input.expand(sample_dim, batch_dim, *self.input_shape)
        if cond_has_sample:
            condition = condition.expand(sample_dim, batch_dim, *self.condition_shape)
        else:
            condition = (
                condition.expand(batch_dim, *self.condition_shape)
                .unsqueeze(0)
                .expand(sample_dim, batch_dim, *self.condition_shape)
            )
        return input, condition, batch_dim


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
        samples = self.sample(sample_shape, condition, **kwargs)
        log_probs = self.log_prob(samples, condition, **kwargs)
        return samples, log_probs
