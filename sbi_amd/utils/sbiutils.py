"""z-scoring and data-hygiene helpers on the NSF/NPE path.

Behavioural mirrors of sbi/utils/sbiutils.py: z_score_parser (:154-189),
z_standardization (:376-415), standardizing_net statistics (:431-488),
handle_invalid_x (:491-525), within_support (:729-766).
"""

from __future__ import annotations

import logging
import warnings
from typing import Callable, Optional, Tuple

import torch
from torch import Tensor


def z_score_parser(z_score_flag: Optional[str] = None) -> Tuple[bool, bool]:
    """-> (do z-score, structured dims)."""
    if z_score_flag is None or z_score_flag == "none":
        return False, False
    if z_score_flag in ("independent", "structured"):
        return True, z_score_flag == "structured"
    if z_score_flag == "transform_to_unconstrained":
        return False, False
    raise ValueError(
        "Invalid z-scoring option. Use 'none', 'independent' 'structured' or "
        "'transform_to_unconstrained'."
    )


def assert_transform_to_unconstrained_supported(z_score_x: Optional[str], builder_name: str, suggestion: str):
    if z_score_x == "transform_to_unconstrained":
        raise ValueError(
            f"`z_score_x='transform_to_unconstrained'` is not supported by `{builder_name}`. {suggestion}"
        )


def handle_invalid_x(x: Tensor, exclude_invalid_x: bool = True) -> Tuple[Tensor, int, int]:
    """Rows with NaN / Inf -> (is_valid mask, num_nans, num_infs) (sbiutils.py:491-525)."""
    batch = x.reshape(x.shape[0], -1)
    x_is_nan = torch.isnan(batch).any(dim=1)
    x_is_inf = torch.isinf(batch).any(dim=1)
    num_nans = int(x_is_nan.sum().item())
    num_infs = int(x_is_inf.sum().item())
    if exclude_invalid_x:
        is_valid = ~x_is_nan & ~x_is_inf
    else:
        is_valid = torch.ones(batch.shape[0], dtype=torch.bool, device=x.device)
    return is_valid, num_nans, num_infs


def warn_on_invalid_x(num_nans: int, num_infs: int, exclude_invalid_x: bool) -> None:
    if num_nans + num_infs > 0:
        if exclude_invalid_x:
            logging.warning(
                f"Found {num_nans} NaN simulations and {num_infs} Inf simulations. They will be excluded "
                "from training."
            )
        else:
            logging.warning(
                f"Found {num_nans} NaN simulations and {num_infs} Inf simulations. Training might fail."
            )


def z_standardization(batch_t: Tensor, structured_dims: bool = False, min_std: float = 1e-14):
    """Mean / std used to z-score theta (sbiutils.py:376-415)."""
    is_valid_t, *_ = handle_invalid_x(batch_t, True)
    t = batch_t[is_valid_t]
    if structured_dims:
        t_mean = torch.mean(t)
        sample_std = torch.std(t, dim=1)
        sample_std[sample_std < min_std] = min_std
        t_std = torch.mean(sample_std)
    else:
        t_mean = torch.mean(t, dim=0)
        t_std = torch.std(t, dim=0)
        t_std[t_std < min_std] = min_std
    return t_mean, t_std


def standardizing_stats(batch_t: Tensor, structured_dims: bool = False, min_std: float = 1e-7):
    """Mean / std of the ``Standardize`` layer prepended to the x embedding
    (sbiutils.py:431-488, incl. the single-row special case)."""
    is_valid_t, *_ = handle_invalid_x(batch_t, True)
    t = batch_t[is_valid_t]
    t_mean = torch.mean(t) if structured_dims else torch.mean(t, dim=0)
    if len(batch_t) > 1:
        if structured_dims:
            sample_std = torch.std(t, dim=1)
            sample_std[sample_std < min_std] = min_std
            t_std = torch.mean(sample_std)
        else:
            t_std = torch.std(t, dim=0)
            t_std[t_std < min_std] = min_std
    else:
        t_std = torch.ones(1)
        logging.warning(
            "Using a one-dimensional batch will instantiate a Standardize transform with (mean, std) "
            "parameters which are not representative of the data."
        )
    if torch.isnan(t_mean).any() or torch.isnan(t_std).any():
        raise AssertionError(
            "Training data mean or std for standardizing net must not contain NaNs. In case you are "
            "encoding missing trials with NaNs, consider setting z_score_x='none' to disable z-scoring."
        )
    return t_mean, t_std


def within_support(distribution, samples: Tensor) -> Tensor:
    """Boolean mask of samples inside the prior support (sbiutils.py:729-766)."""
    try:
        sample_check = distribution.support.check(samples)
        if sample_check.shape == samples.shape:
            sample_check = torch.all(sample_check, dim=-1)
        return sample_check
    except (NotImplementedError, AttributeError):
        return torch.isfinite(distribution.log_prob(samples))


def warn_if_outside_prior_support(prior, samples: Tensor) -> None:
    inside = within_support(prior, samples)
    if not bool(inside.all()):
        frac = 1.0 - float(inside.float().mean())
        warnings.warn(
            f"{frac:.1%} of the samples drawn without rejection lie outside the prior support.",
            stacklevel=2,
        )


def mcmc_transform(prior, num_prior_samples_for_zscoring: int = 1000, enable_transform: bool = True,
                   device="cpu", **kwargs):
    """Transform applied to parameters during MCMC (sbiutils.py:867-980): bounded supports are mapped to
    unbounded space with `biject_to`, unbounded ones are z-scored with the prior's mean / std.  The returned
    transform's forward maps constrained -> unconstrained."""
    import torch.distributions.transforms as torch_tf
    from torch.distributions import biject_to, constraints

    if enable_transform:
        def prior_mean_std_transform():
            try:
                prior_mean, prior_std = prior.mean.to(device), prior.stddev.to(device)
            except (NotImplementedError, AttributeError):
                warnings.warn("The passed prior has no mean or stddev attribute, estimating them from samples to "
                              "build affine standardizing transform.", stacklevel=2)
                th = prior.sample(torch.Size((num_prior_samples_for_zscoring,)))
                prior_mean, prior_std = th.mean(dim=0).to(device), th.std(dim=0).to(device)
            return torch_tf.AffineTransform(loc=prior_mean, scale=prior_std)

        try:
            _ = prior.support
            has_support = True
        except (NotImplementedError, AttributeError):
            warnings.warn("The passed prior has no support property, transform will be constructed from mean and "
                          "std. If the passed prior is supposed to be bounded consider implementing the "
                          "prior.support property.", stacklevel=2)
            has_support = False
        if has_support:
            constraint = prior.support.base_constraint if hasattr(prior.support, "base_constraint") else prior.support
            if getattr(prior.support, "is_discrete", False) or isinstance(constraint, constraints._Real):
                transform = prior_mean_std_transform()
            else:
                transform = biject_to(prior.support)
        else:
            transform = prior_mean_std_transform()
    else:
        transform = torch_tf.identity_transform
    if not isinstance(transform, torch_tf.IndependentTransform):
        transform = torch_tf.IndependentTransform(transform, reinterpreted_batch_ndims=1)
    check_transform(prior, transform)
    return transform.inv


def check_transform(prior, transform, atol: float = 1e-3) -> None:
    """sbiutils.py:983-1003."""
    try:
        theta = prior.sample(torch.Size((2,)))
    except NotImplementedError:
        theta = prior.mean.repeat(2, *[1] * prior.mean.dim())
    theta_unconstrained = transform.inv(theta)
    assert theta_unconstrained.shape == theta.shape, (
        "Mismatch between transformed and untransformed space. Note that you cannot use a transforms when using a "
        "MultipleIndependent prior with a Dirichlet prior.")
    assert torch.allclose(theta, transform(theta_unconstrained), atol=atol), \
        "Original and re-transformed parameters must be close to each other."


def gradient_ascent(potential_fn: Callable, inits: Tensor, theta_transform=None, num_iter: int = 1_000,
                    num_to_optimize: int = 100, learning_rate: float = 0.01, save_best_every: int = 10,
                    show_progress_bars: bool = False, interruption_note: str = "") -> Tuple[Tensor, Tensor]:
    """`argmax` and `max` of `potential_fn` by Adam ascent in the unconstrained space of `theta_transform`, started
    from the `num_to_optimize` best of `inits` (sbi/utils/sbiutils.py:1160-1286: same selection, optimizer, the
    best-so-far bookkeeping every `save_best_every` iterations, Ctrl-C returns the current best).  With an NSF
    estimator on a ROCm device every evaluation is one launch of the batched log_prob kernel and every gradient the
    fused backward pass (d log_prob / d theta)."""
    import torch.distributions.transforms as torch_tf

    if theta_transform is None:
        theta_transform = torch_tf.IndependentTransform(torch_tf.identity_transform, reinterpreted_batch_ndims=1)
    init_probs = potential_fn(inits).detach()
    inits = inits.to(init_probs.device)
    sort_indices = torch.argsort(init_probs, dim=0)
    sorted_inits = inits[sort_indices]
    optimize_inits = sorted_inits[-num_to_optimize:]
    best_log_prob_iter = torch.max(init_probs)
    best_theta_iter = sorted_inits[-1]
    best_theta_overall = best_theta_iter.detach().clone()
    best_log_prob_overall = best_log_prob_iter.detach().clone()
    argmax_, max_val = best_theta_overall, best_log_prob_overall
    # NOTE (as in the reference): `best_theta_overall` starts in constrained space and is replaced by
    # unconstrained-space points once an iteration improves on it; the return maps it back with `.inv`
    optimize_inits = theta_transform(optimize_inits).detach().clone()
    optimize_inits.requires_grad_(True)
    optimizer = torch.optim.Adam([optimize_inits], lr=learning_rate)
    iter_ = 0
    try:
        while iter_ < num_iter:
            optimizer.zero_grad()
            probs = potential_fn(theta_transform.inv(optimize_inits)).squeeze()
            (-probs.sum()).backward()
            optimizer.step()
            with torch.no_grad():
                if iter_ % save_best_every == 0 or iter_ == num_iter - 1:
                    log_probs_of_optimized = potential_fn(theta_transform.inv(optimize_inits))
                    best_theta_iter = optimize_inits[torch.argmax(log_probs_of_optimized)].unsqueeze(0)
                    best_log_prob_iter = potential_fn(theta_transform.inv(best_theta_iter))
                    if best_log_prob_iter > best_log_prob_overall:
                        best_theta_overall = best_theta_iter.detach().clone()
                        best_log_prob_overall = best_log_prob_iter.detach().clone()
                if show_progress_bars:
                    print("\r", f"Optimizing MAP estimate. Iterations: {iter_ + 1} / {num_iter}. Performance in "
                          f"iteration {divmod(iter_ + 1, save_best_every)[0] * save_best_every}: "
                          f"{best_log_prob_iter.item():.2f} (= unnormalized log-prob). Press Ctrl-C to interrupt.",
                          end="")
                argmax_ = theta_transform.inv(best_theta_overall)
                max_val = best_log_prob_overall
            iter_ += 1
    except KeyboardInterrupt:
        print(f"Optimization was interrupted after {iter_} iterations. " + interruption_note)
        return argmax_, max_val
    return theta_transform.inv(best_theta_overall), max_val
