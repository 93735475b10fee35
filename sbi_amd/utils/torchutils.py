"""Tensor / device helpers on the NSF/NPE path.

Mirrors of sbi/utils/torchutils.py: process_device (:54-103), BoxUniform
(:525-650), create_alternating_binary_mask (:396-410), repeat_rows (:314-330),
searchsorted (:449-463), ensure_theta_batched / atleast_2d.
"""

from __future__ import annotations

import warnings
from typing import Optional, Union

import torch
from torch import Tensor
from torch.distributions import Independent, Uniform


def process_device(device: Union[str, torch.device]) -> str:
    """Resolve 'cpu' / 'gpu' / 'cuda' / 'cuda:i' (ROCm devices are 'cuda' in PyTorch)."""
    if isinstance(device, torch.device):
        device = str(device)
    if device == "cpu":
        return "cpu"
    if device == "gpu":
        if not torch.cuda.is_available():
            raise RuntimeError("Neither a ROCm/CUDA device is available; use device='cpu'.")
        device = "cuda"
    try:
        dev = torch.device(device)
    except RuntimeError as e:
        raise RuntimeError(f"Could not instantiate torch.device('{device}')") from e
    if dev.type == "cuda":
        if not torch.cuda.is_available():
            raise RuntimeError(f"Device {device} requested but no ROCm device is visible.")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        return f"cuda:{idx}"
    return str(dev)


def check_if_prior_on_device(device: Union[str, torch.device], prior=None) -> None:
    if prior is None:
        return
    prior_device = prior.sample((1,)).device
    if torch.device(device).type != prior_device.type:
        raise AssertionError(
            f"Prior device '{prior_device}' must match training device '{device}'. When training on GPU "
            "make sure to pass a prior initialized on the GPU as well, e.g., "
            "`prior = torch.distributions.Normal(torch.zeros(2, device='cuda'), scale=1.0)`."
        )


def atleast_2d(*arys: Tensor):
    res = [a.unsqueeze(0) if a.dim() < 2 else a for a in arys]
    return res[0] if len(res) == 1 else res


def ensure_theta_batched(theta: Tensor) -> Tensor:
    return theta.unsqueeze(0) if theta.dim() == 1 else theta


def create_alternating_binary_mask(features: int, even: bool = True) -> Tensor:
    mask = torch.zeros(features).byte()
    mask[(0 if even else 1) :: 2] += 1
    return mask


def repeat_rows(x: Tensor, num_reps: int) -> Tensor:
    shape = x.shape
    return x.unsqueeze(1).expand(shape[0], num_reps, *shape[1:]).reshape(-1, *shape[1:])


def searchsorted(bin_locations: Tensor, inputs: Tensor, eps: float = 1e-6) -> Tensor:
    """Bin index of each input; bumps the last edge IN PLACE like the reference."""
    bin_locations[..., -1] += eps
    return torch.sum(inputs[..., None] >= bin_locations, dim=-1) - 1


class BoxUniform(Independent):
    """Uniform box prior with event dimension and device handling (torchutils.py:525-650)."""

    def __init__(self, low, high, reinterpreted_batch_ndims: int = 1, device: Optional[str] = None):
        if device is None:
            device = low.device.type if isinstance(low, Tensor) else "cpu"
        self.device = device
        self.low = torch.as_tensor(low, dtype=torch.float32, device=device)
        self.high = torch.as_tensor(high, dtype=torch.float32, device=device)
        if isinstance(low, Tensor) and isinstance(high, Tensor) and low.device != high.device:
            raise RuntimeError("Expected all tensors to be on the same device")
        super().__init__(Uniform(low=self.low, high=self.high, validate_args=False),
                         reinterpreted_batch_ndims, validate_args=False)

    def to(self, device: Union[str, torch.device]) -> "BoxUniform":
        self.device = str(device)
        self.low = self.low.to(device)
        self.high = self.high.to(device)
        super().__init__(Uniform(low=self.low, high=self.high, validate_args=False),
                         self.reinterpreted_batch_ndims, validate_args=False)
        return self
