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
    """-> (mask of rows to keep, number of rows with a NaN, number of rows with an Inf); behaviour of
    sbiutils.py:491-525.  One pass over the flattened rows; both counts leave the device in one read."""
    rows = x.reshape(len(x), -1)
    has_nan = rows.isnan().any(1)
    has_inf = rows.isinf().any(1)
    counts = torch.stack([has_nan.sum(), has_inf.sum()]).tolist()
    keep = ~(has_nan | has_inf) if exclude_invalid_x else torch.ones_like(has_nan)
    return keep, int(counts[0]), int(counts[1])


def warn_on_invalid_x(num_nans: int, num_infs: int, exclude_invalid_x: bool) -> None:
    if num_nans == 0 and num_infs == 0:
        return
    consequence = "They will be excluded from training." if exclude_invalid_x else "Training might fail."
    logging.warning(f"Found {num_nans} NaN simulations and {num_infs} Inf simulations. {consequence}")


def _location_scale(rows: Tensor, structured: bool, floor: float):
    """Per-dimension (independent) or single (structured: one mean over everything, the average of the per-row
    standard deviations) location / scale of the finite rows, scale floored at `floor`."""
    if structured:
        return rows.mean(), torch.clamp_min(rows.std(dim=1), floor).mean()
    return rows.mean(dim=0), torch.clamp_min(rows.std(dim=0), floor)


def z_standardization(batch_t: Tensor, structured_dims: bool = False, min_std: float = 1e-14):
    """Mean / std used to z-score theta (sbiutils.py:376-415; golden: tests/golden/reference_intree.pt)."""
    keep = handle_invalid_x(batch_t, True)[0]
    return _location_scale(batch_t[keep], structured_dims, min_std)


def standardizing_stats(batch_t: Tensor, structured_dims: bool = False, min_std: float = 1e-7):
    """Mean / std of the ``Standardize`` layer in front of the x embedding (sbiutils.py:431-488; a single-row
    batch gets std 1 and a warning, NaN statistics are refused)."""
    keep = handle_invalid_x(batch_t, True)[0]
    if len(batch_t) > 1:
        t_mean, t_std = _location_scale(batch_t[keep], structured_dims, min_std)
    else:
        kept = batch_t[keep]
        t_mean = kept.mean() if structured_dims else kept.mean(dim=0)      # (no std of a single row: torch warns, and
        t_std = torch.ones(1)                                              #  the reference substitutes 1 anyway)
        logging.warning("Using a one-dimensional batch will instantiate a Standardize transform with (mean, std) "
                        "parameters which are not representative of the data.")
    if bool(torch.isnan(t_mean).any() | torch.isnan(t_std).any()):
        raise AssertionError("Training data mean or std for standardizing net must not contain NaNs. In case you "
                             "are encoding missing trials with NaNs, consider setting z_score_x='none' to disable "
                             "z-scoring.")
    return t_mean, t_std


def within_support(distribution, samples: Tensor) -> Tensor:
    """Boolean mask of the rows of `samples` inside `distribution`'s support (sbiutils.py:729-766): the support's
    own check, reduced over the event dimension when it answers per coordinate; distributions without a usable
    support fall back to "log_prob is finite"."""
    try:
        inside = distribution.support.check(samples)
    except (NotImplementedError, AttributeError):
        return distribution.log_prob(samples).isfinite()
    return inside.all(dim=-1) if inside.shape == samples.shape else inside


def warn_if_outside_prior_support(prior, samples: Tensor) -> None:
    inside = within_support(prior, samples)
    if not bool(inside.all()):
        frac = 1.0 - float(inside.float().mean())
        warnings.warn(
            f"{frac:.1%} of the samples drawn without rejection lie outside the prior support.",
            stacklevel=2,
        )


class _Standardise:
    """Builds the affine map used for unbounded priors: location / scale from the prior's moments, or from
    `n_draws` prior draws when the prior does not publish them."""

    def __init__(self, prior, n_draws: int, device):
        self.prior, self.n_draws, self.device = prior, n_draws, device

    def moments(self):
        try:
            return self.prior.mean.to(self.device), self.prior.stddev.to(self.device)
        except (NotImplementedError, AttributeError):
            warnings.warn("The passed prior has no mean or stddev attribute, estimating them from samples to "
                          "build affine standardizing transform.", stacklevel=3)
            draws = self.prior.sample(torch.Size((self.n_draws,)))
            return draws.mean(dim=0).to(self.device), draws.std(dim=0).to(self.device)

    def __call__(self):
        from torch.distributions.transforms import AffineTransform

        loc, scale = self.moments()
        return AffineTransform(loc=loc, scale=scale)


def _support_or_none(prior):
    try:
        return prior.support
    except (NotImplementedError, AttributeError):
        warnings.warn("The passed prior has no support property, transform will be constructed from mean and "
                      "std. If the passed prior is supposed to be bounded consider implementing the "
                      "prior.support property.", stacklevel=3)
        return None


def mcmc_transform(prior, num_prior_samples_for_zscoring: int = 1000, enable_transform: bool = True,
                   device="cpu", **kwargs):
    """The parameter transform MCMC / MAP / rejection sampling work in (behaviour of sbi/utils/sbiutils.py:867-980,
    pinned by tests/golden/mcmc_reference.pt and tests/test_mcmc_cpu.py).  Decision table:

    ==============================================  =========================================
    prior                                            constrained -> unconstrained map
    ==============================================  =========================================
    `enable_transform=False`                         identity
    no `.support`, real or discrete support          z-scoring with the prior's mean / std
    any other (bounded) support                      `biject_to(support)` inverted
    ==============================================  =========================================

    The result always has event dim 1 and its FORWARD direction maps constrained -> unconstrained."""
    from torch.distributions import biject_to, constraints
    from torch.distributions.transforms import IndependentTransform, identity_transform

    to_constrained = identity_transform
    if enable_transform:
        support = _support_or_none(prior)
        bounded = False
        if support is not None:
            innermost = getattr(support, "base_constraint", support)
            bounded = not (getattr(support, "is_discrete", False) or isinstance(innermost, constraints._Real))
        to_constrained = biject_to(support) if bounded else _Standardise(prior, num_prior_samples_for_zscoring, device)()
    if not isinstance(to_constrained, IndependentTransform):
        to_constrained = IndependentTransform(to_constrained, reinterpreted_batch_ndims=1)
    check_transform(prior, to_constrained)
    return to_constrained.inv


def check_transform(prior, transform, atol: float = 1e-3) -> None:
    """Round trip of two prior points through `transform` (unconstrained -> constrained direction) must keep the
    shape and reproduce the points (sbiutils.py:983-1003)."""
    try:
        probe = prior.sample(torch.Size((2,)))
    except NotImplementedError:
        probe = torch.stack([prior.mean, prior.mean])
    back = transform.inv(probe)
    if back.shape != probe.shape:
        raise AssertionError("Mismatch between transformed and untransformed space. Note that you cannot use a "
                             "transforms when using a MultipleIndependent prior with a Dirichlet prior.")
    if not torch.allclose(probe, transform(back), atol=atol):
        raise AssertionError("Original and re-transformed parameters must be close to each other.")


class _Incumbent:
    """Best point seen so far, kept in CONSTRAINED coordinates together with its potential value."""

    def __init__(self, theta: Tensor, value: Tensor):
        self.theta, self.value = theta.detach().clone(), value.detach().clone()

    def offer(self, theta: Tensor, value: Tensor) -> None:
        if bool(value > self.value):
            self.theta, self.value = theta.detach().clone(), value.detach().clone()


def gradient_ascent(potential_fn: Callable, inits: Tensor, theta_transform=None, num_iter: int = 1_000,
                    num_to_optimize: int = 100, learning_rate: float = 0.01, save_best_every: int = 10,
                    show_progress_bars: bool = False, interruption_note: str = "") -> Tuple[Tensor, Tensor]:
    """(argmax, max) of `potential_fn` by Adam ascent (behaviour of sbi/utils/sbiutils.py:1160-1286, pinned by
    tests/golden/rejection_reference.pt): the `num_to_optimize` highest-potential rows of `inits` are moved to the
    unconstrained side of `theta_transform` and ascended jointly (the summed potential is the objective, so the rows
    do not interact); after iterations 0, `save_best_every`, 2 `save_best_every`, ... and after the last one the
    current front-runner challenges the incumbent.  Ctrl-C returns the incumbent.  One deliberate difference: the
    incumbent is stored in constrained coordinates throughout, so when no iteration ever beats the best initial
    point that point is returned as it is (the reference would push it through the inverse transform a second
    time).  With an NSF estimator on a ROCm device every evaluation is one launch of the batched log_prob kernel and
    every gradient one fused backward pass (d log_prob / d theta)."""
    if theta_transform is None:
        to_constrained = lambda u: u     # noqa: E731
        to_unconstrained = to_constrained
    else:
        to_constrained, to_unconstrained = theta_transform.inv, theta_transform

    start_values = potential_fn(inits).detach()
    inits = inits.to(start_values.device)
    ranking = torch.argsort(start_values, dim=0)                   # ascending, as the golden run's tie order
    incumbent = _Incumbent(inits[ranking[-1]], start_values.max())
    points = to_unconstrained(inits[ranking[-num_to_optimize:]]).detach().clone().requires_grad_(True)
    adam = torch.optim.Adam([points], lr=learning_rate)
    checkpoints = set(range(0, num_iter, save_best_every)) | {num_iter - 1}
    done = 0
    try:
        for done in range(num_iter):
            adam.zero_grad()
            objective = potential_fn(to_constrained(points)).squeeze().sum()
            (-objective).backward()
            adam.step()
            if done in checkpoints:
                with torch.no_grad():
                    values = potential_fn(to_constrained(points))
                    leader = to_constrained(points[torch.argmax(values)].unsqueeze(0))
                    incumbent.offer(leader, potential_fn(leader))
            if show_progress_bars:
                print(f"\rMAP search {done + 1}/{num_iter}: best unnormalized log-prob {float(incumbent.value):.2f} "
                      "(Ctrl-C stops and keeps it)", end="")
    except KeyboardInterrupt:
        print(f"Optimization was interrupted after {done} iterations. " + interruption_note)
    return incumbent.theta, incumbent.value
