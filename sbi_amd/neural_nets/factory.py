"""``posterior_nn`` factory for the accelerated path.

Same call signature and behaviours as sbi/neural_nets/factory.py:323-430 for
``model="nsf"``: returns ``build_fn(batch_theta, batch_x)``; unknown kwargs warn
and are forwarded (factory_config_test.py:68-70); other model names are accepted
at factory time and raise ``NotImplementedError`` when the net is built
(:179-183) -- they are other estimator families, outside this path.
"""

from __future__ import annotations

import warnings
from typing import Any, Callable, Optional

from torch import Tensor, nn

from sbi_amd.neural_nets.net_builders.estimator_configs import MAFRQSConfig, NSFConfig, ZukoNSFConfig

_NSF_FIELDS = {"hidden_features", "num_transforms", "num_bins", "num_blocks", "dropout_probability",
               "use_batch_norm", "tail_bound", "hidden_layers_spline_context", "dtype"}
_MAF_RQS_FIELDS = (_NSF_FIELDS - {"hidden_layers_spline_context"}) | {"tails", "min_bin_width", "min_bin_height",
                                                                        "min_derivative"}
_MODELS = {"nsf": (NSFConfig, _NSF_FIELDS), "maf_rqs": (MAFRQSConfig, _MAF_RQS_FIELDS)}


def posterior_nn(
    model: str = "nsf",
    z_score_theta: Optional[str] = "independent",
    z_score_x: Optional[str] = "independent",
    hidden_features: int = 50,
    num_transforms: int = 5,
    num_bins: int = 10,
    embedding_net: nn.Module = nn.Identity(),
    **kwargs: Any,
) -> Callable[[Tensor, Tensor], nn.Module]:
    """Return a function that builds the posterior density estimator from (theta, x) batches."""
    model_fields = _MODELS.get(model, _MODELS["nsf"])[1]
    known = {k: v for k, v in kwargs.items() if k in model_fields}
    unknown = {k: v for k, v in kwargs.items() if k not in model_fields}
    if unknown:
        warnings.warn(f"Unknown kwargs {sorted(unknown)} are forwarded to the builder.", UserWarning, stacklevel=2)

    def build_fn(batch_theta: Tensor, batch_x: Tensor):
        if model == "zuko_nsf":
            return ZukoNSFConfig(z_score_input=z_score_theta, z_score_condition=z_score_x,
                                 embedding_net=None if isinstance(embedding_net, nn.Identity) else embedding_net,
                                 hidden_features=hidden_features, num_transforms=num_transforms, num_bins=num_bins,
                                 extra_kwargs={**known, **unknown}).build(batch_theta, batch_x)
        if model not in _MODELS:
            raise NotImplementedError(
                f"sbi_amd implements the 'nsf', 'maf_rqs' and 'zuko_nsf' posterior estimators (got model={model!r}); other "
                "model families are outside the accelerated path."
            )
        cfg = _MODELS[model][0](
            z_score_input=z_score_theta, z_score_condition=z_score_x,
            embedding_net=None if isinstance(embedding_net, nn.Identity) else embedding_net,
            hidden_features=hidden_features, num_transforms=num_transforms, num_bins=num_bins,
            extra_kwargs=unknown, **known,
        )
        return cfg.build(batch_theta, batch_x)

    return build_fn


def posterior_flow_nn(*args, **kwargs):
    """``sbi.neural_nets.posterior_flow_nn`` (factory.py:531-620) for the FMPE path; see
    ``sbi_amd.inference.trainers.vfpe.fmpe.posterior_flow_nn`` (imported lazily: it needs the trainer package)."""
    from sbi_amd.inference.trainers.vfpe.fmpe import posterior_flow_nn as _impl

    return _impl(*args, **kwargs)
