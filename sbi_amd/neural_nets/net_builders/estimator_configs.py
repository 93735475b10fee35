"""Typed configuration of the NSF density estimator.

``NSFConfig`` mirrors sbi/neural_nets/net_builders/estimator_configs.py:1223-1237
(fields inherited from the flow base classes :930-968, :1172-1189): a frozen
dataclass whose ``build(batch_input, batch_condition)`` returns the estimator,
with the same field names, defaults, ``extra_kwargs`` escape valve and
validation of the z-score literals.
"""

from __future__ import annotations

from dataclasses import dataclass, field, fields
from typing import Any, Dict, Optional

import torch
from torch import Tensor, nn

_Z_SCORE_VALUES = ("none", "independent", "structured", "transform_to_unconstrained")


@dataclass(frozen=True)
class NSFConfig:
    z_score_input: Optional[str] = "independent"
    z_score_condition: Optional[str] = "independent"
    embedding_net: Optional[nn.Module] = None
    hidden_features: int = 50
    num_transforms: int = 5
    num_blocks: int = 2
    dropout_probability: float = 0.0
    use_batch_norm: bool = False
    dtype: torch.dtype = torch.float32
    num_bins: int = 10
    tail_bound: float = 3.0
    hidden_layers_spline_context: int = 1
    extra_kwargs: Dict[str, Any] = field(default_factory=dict)

    def __post_init__(self):
        for name in ("z_score_input", "z_score_condition"):
            v = getattr(self, name)
            if v is None:
                object.__setattr__(self, name, "none")
            elif v not in _Z_SCORE_VALUES:
                raise ValueError(f"{name} must be one of {_Z_SCORE_VALUES} or None, got {v!r}")
        for name in ("hidden_features", "num_transforms", "num_blocks", "num_bins"):
            if int(getattr(self, name)) < 1:
                raise ValueError(f"{name} must be a positive integer")

    def _build_kwargs(self) -> Dict[str, Any]:
        kw = {f.name: getattr(self, f.name) for f in fields(self) if f.name != "extra_kwargs"}
        kw["z_score_x"] = kw.pop("z_score_input")
        kw["z_score_y"] = kw.pop("z_score_condition")
        if kw["embedding_net"] is None:
            kw["embedding_net"] = nn.Identity()
        kw.update(self.extra_kwargs)
        return kw

    def build(self, batch_input: Tensor, batch_condition: Tensor):
        from sbi_amd.neural_nets.net_builders.flow import build_nsf

        return build_nsf(batch_x=batch_input, batch_y=batch_condition, **self._build_kwargs())

    def __repr__(self) -> str:   # only non-default fields, like the reference's configs
        parts = []
        for f in fields(self):
            v = getattr(self, f.name)
            default = f.default if f.default is not field else None
            if f.name == "extra_kwargs":
                if v:
                    parts.append(f"extra_kwargs={v!r}")
            elif v != default:
                parts.append(f"{f.name}={v!r}")
        return f"{type(self).__name__}({', '.join(parts)})"


@dataclass(frozen=True, repr=False)
class MAFRQSConfig(NSFConfig):
    """Mirror of sbi's ``MAFRQSConfig`` (estimator_configs.py:1200-1219): the flow-base fields plus the spline's."""

    tails: Optional[str] = "linear"
    min_bin_width: float = 1e-3
    min_bin_height: float = 1e-3
    min_derivative: float = 1e-3

    def _build_kwargs(self) -> Dict[str, Any]:
        kw = super()._build_kwargs()
        kw.pop("hidden_layers_spline_context", None)   # an NSF-only field
        return kw

    def build(self, batch_input: Tensor, batch_condition: Tensor):
        from sbi_amd.neural_nets.net_builders.flow import build_maf_rqs

        return build_maf_rqs(batch_x=batch_input, batch_y=batch_condition, **self._build_kwargs())


@dataclass(frozen=True)
class ZukoNSFConfig:
    """Mirror of sbi's ``ZukoNSFConfig`` (estimator_configs.py: zuko flow base fields + ``num_bins``)."""

    z_score_input: Optional[str] = "independent"
    z_score_condition: Optional[str] = "independent"
    embedding_net: Optional[nn.Module] = None
    hidden_features: Any = 50
    num_transforms: int = 5
    num_bins: int = 10
    extra_kwargs: Dict[str, Any] = field(default_factory=dict)

    def __post_init__(self):
        for name in ("z_score_input", "z_score_condition"):
            v = getattr(self, name)
            if v is None:
                object.__setattr__(self, name, "none")
            elif v not in _Z_SCORE_VALUES:
                raise ValueError(f"{name} must be one of {_Z_SCORE_VALUES} or None, got {v!r}")

    def build(self, batch_input: Tensor, batch_condition: Tensor):
        from sbi_amd.neural_nets.net_builders.flow import build_zuko_nsf

        return build_zuko_nsf(batch_x=batch_input, batch_y=batch_condition, z_score_x=self.z_score_input,
                              z_score_y=self.z_score_condition, hidden_features=self.hidden_features,
                              num_transforms=self.num_transforms, num_bins=self.num_bins,
                              embedding_net=nn.Identity() if self.embedding_net is None else self.embedding_net,
                              **self.extra_kwargs)
