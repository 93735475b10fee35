"""Accept/reject sampling from the estimator inside the prior support.

Same contract as sbi/samplers/rejection/rejection.py:230-457
(``accept_reject_sample``): draws ``sampling_batch_size`` candidates from
``proposal``, keeps those ``accept_reject_fn`` accepts, adapts the batch size
from the running acceptance rate (:406-409), warns once below
``warn_acceptance``, honours ``max_sampling_time`` / ``return_partial_on_timeout``
and returns ``(samples (num_samples, num_xos, *event), acceptance_rate (num_xos,))``.

MI355X-first differences (results are the same set of samples in the same order):
on a ROCm device an iteration is the proposal's kernels plus ONE launch (csrc/compact.hip:
acceptance test -- fused for a box prior, else a mask from `accept_reject_fn` -- single-pass
scan, order-preserving scatter into a preallocated buffer) and ONE small read-back (the
per-condition accept counts the batch-size rule needs).  Host tensors (the CPU golden replays)
take the same steps as torch operations.
"""

from __future__ import annotations

import logging
import time
import warnings
from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor


@torch.no_grad()
def accept_reject_sample(
    proposal: Callable,
    accept_reject_fn: Callable[[Tensor], Tensor],
    num_samples: int,
    num_xos: int = 1,
    show_progress_bars: bool = False,
    warn_acceptance: float = 0.01,
    sample_for_correction_factor: bool = False,
    max_sampling_batch_size: int = 10_000,
    proposal_sampling_kwargs: Optional[Dict] = None,
    alternative_method: Optional[str] = None,
    max_sampling_time: Optional[float] = None,
    return_partial_on_timeout: bool = False,
    acceptance_on_device: bool = True,
    **kwargs,
) -> Tuple[Tensor, Tensor]:
    """(sbi/samplers/rejection/rejection.py:230-457; `acceptance_on_device=False` leaves the returned acceptance rate on
    the host, where it is computed: callers that ignore it save a copy.)"""
    if kwargs:
        logging.warning(
            "You passed arguments to `rejection_sampling_parameters` that are unused when you do not "
            f"specify a `proposal` in the same dictionary. The unused arguments are: {kwargs}"
        )
    if proposal_sampling_kwargs is None:
        proposal_sampling_kwargs = {}
    if "condition" in proposal_sampling_kwargs:
        num_xos = proposal_sampling_kwargs["condition"].shape[0]

    out: Optional[Tensor] = None            # (num_samples, num_xos, *event)
    filled: Optional[Tensor] = None         # (num_xos,) int64, on device
    acceptance_rate = torch.full((num_xos,), float("nan"))
    num_remaining = num_samples
    sampling_batch_size = min(num_samples, max_sampling_batch_size)
    num_samples_possible = 0
    leakage_warning_raised = False
    start_time = time.time()
    candidates = None

    while num_remaining > 0:
        if max_sampling_time is not None and (time.time() - start_time) > max_sampling_time:
            num_collected = 0 if filled is None else int(filled.min().item())
            if return_partial_on_timeout and num_collected > 0:
                warnings.warn(
                    f"Timeout exceeded after collecting {num_collected}/{num_samples} samples. "
                    "Returning partial results.", stacklevel=2,
                )
                return out[:num_collected], acceptance_rate.to(out.device)
            raise RuntimeError(
                "Sampling aborted early because rejection sampling exceeded max_sampling_time. This is "
                "likely due to extremely low acceptance. You can disable rejection sampling using "
                "`reject_outside_prior=False` to draw samples directly from the trained estimator. "
                "Consider switching to MCMC or VI, or checking for model misspecification."
            )

        candidates = proposal(torch.Size((sampling_batch_size,)), **proposal_sampling_kwargs)
        cand = candidates.reshape(sampling_batch_size, num_xos, *candidates.shape[candidates.ndim - 1 :])
        on_device = cand.is_cuda and cand.dtype == torch.float32 and cand.ndim == 3
        box = getattr(accept_reject_fn, "box_bounds", None) if on_device else None
        if box is not None and (box[0].numel() != cand.shape[-1] or box[0].device != cand.device):
            box = None
        if on_device:
            # ---- one launch: acceptance + stable compaction + running counts (include/sbi_amd_nsf.h)
            from sbi_amd import _lib

            lib = _lib.load()
            if out is None:
                ev = cand.shape[-1]
                out = torch.empty((num_samples, num_xos, ev), dtype=cand.dtype, device=cand.device)
                words = lib.sbi_amd_accept_compact_scan_words(max(max_sampling_batch_size, sampling_batch_size), num_xos)
                zeros = torch.zeros(4 * num_xos + int(words), dtype=torch.long, device=cand.device)    # (one fill)
                state = zeros[: 3 * num_xos]
                control = zeros[3 * num_xos : 4 * num_xos].view(torch.int32)
                scan = zeros[4 * num_xos :]
                generation = 0
                filled = state[:num_xos]
            mask = None
            if box is None:
                mask = accept_reject_fn(candidates).reshape(sampling_batch_size, num_xos).to(torch.bool).contiguous()
            cand_c = cand.contiguous()
            generation += 1
            with torch.cuda.device(cand.device):
                rc = lib.sbi_amd_accept_compact(
                    _lib.ptr(cand_c), _lib.ptr(mask), None if box is None else _lib.ptr(box[0]),
                    None if box is None else _lib.ptr(box[1]), sampling_batch_size, num_xos, cand.shape[-1], _lib.ptr(out),
                    num_samples, _lib.ptr(state), _lib.ptr(control), _lib.ptr(scan), generation,
                    _lib.current_stream(cand.device))
            _lib.check(rc, "accept_compact")
            num_samples_possible += sampling_batch_size
            stats = state[num_xos:].cpu().reshape(2, num_xos)[[1, 0]]      # the ONE host read-back: [this call, so far]
        else:
            are_accepted = accept_reject_fn(candidates).reshape(sampling_batch_size, num_xos)
            acc_i = are_accepted.to(torch.long)
            num_accepted = acc_i.sum(dim=0)
            if out is None:
                # one extra row: the dump slot rejected / surplus candidates are scattered to
                buf = torch.empty((num_samples + 1, num_xos, *cand.shape[2:]), dtype=cand.dtype, device=cand.device)
                out = buf[:num_samples]
                flat = buf.view(-1, *cand.shape[2:])
                filled = torch.zeros(num_xos, dtype=torch.long, device=cand.device)
                total_accepted = torch.zeros(num_xos, dtype=torch.long, device=cand.device)
                xo_idx = torch.arange(num_xos, device=cand.device).unsqueeze(0)
            # stable compaction: destination row of every accepted candidate, per condition
            dest = torch.cumsum(acc_i, dim=0) - 1 + filled.unsqueeze(0)              # (bs, num_xos)
            keep = are_accepted & (dest < num_samples)
            dest = torch.where(keep, dest, torch.full_like(dest, num_samples))      # overflow row
            flat.index_copy_(0, (dest * num_xos + xo_idx).reshape(-1), cand.reshape(-1, *cand.shape[2:]))
            filled = torch.clamp(filled + num_accepted, max=num_samples)
            total_accepted += num_accepted
            num_samples_possible += sampling_batch_size
            stats = torch.stack([num_accepted, total_accepted]).cpu()
        min_num_accepted = int(stats[0].min())
        num_remaining -= min_num_accepted
        acceptance_rate = stats[1].to(torch.float32) / num_samples_possible
        min_acceptance_rate = float(acceptance_rate.min())

        sampling_batch_size = min(
            max_sampling_batch_size,
            max(int(1.5 * num_remaining / max(min_acceptance_rate, 1e-12)), 100),
        )
        if (
            num_samples_possible > (sampling_batch_size - 1)
            and min_acceptance_rate < warn_acceptance
            and not leakage_warning_raised
        ):
            if sample_for_correction_factor:
                logging.warning(
                    f"Drawing samples from posterior to estimate the normalizing constant for `log_prob()`. "
                    f"However, only {min_acceptance_rate:.3%} posterior samples are within the prior support "
                    f"(for condition {int(acceptance_rate.argmin())}). It may take a long time to collect the "
                    f"remaining {num_remaining} samples. Consider `log_prob(..., norm_posterior=False)`."
                )
            else:
                msg = (
                    f"Only {min_acceptance_rate:.3%} proposal samples are accepted. It may take a long time "
                    f"to collect the remaining {num_remaining} samples. You can prevent very long runtimes by "
                    "setting `max_sampling_time`, or disabling rejection sampling "
                    "(`reject_outside_prior=False`)."
                )
                if alternative_method is not None:
                    msg += f" Alternatively, consider switching to `{alternative_method}`."
                logging.warning(msg)
            leakage_warning_raised = True

    assert out is not None
    samples = out.reshape(num_samples, *candidates.shape[1:])
    return samples, acceptance_rate.to(samples.device) if acceptance_on_device else acceptance_rate


def rejection_sample(
    potential_fn: Callable,
    proposal,
    theta_transform=None,
    num_samples: int = 1,
    show_progress_bars: bool = False,
    warn_acceptance: float = 0.01,
    max_sampling_batch_size: int = 10_000,
    num_samples_to_find_max: int = 10_000,
    num_iter_to_find_max: int = 100,
    m: float = 1.2,
    max_sampling_time: Optional[float] = None,
    return_partial_on_timeout: bool = False,
    device: str = "cpu",
) -> Tuple[Tensor, Tensor]:
    r"""Rejection sampling of `exp(potential_fn)` with candidates from `proposal`
    (sbi/samplers/rejection/rejection.py:18-227): the envelope is `proposal * M` with
    `log M = max_theta [potential(theta) - proposal.log_prob(theta)] + log m`, the maximum found by gradient
    ascent from the best tenth of `num_samples_to_find_max` proposal draws; a candidate is kept when
    `exp(potential - proposal.log_prob - log M) > u`, `u ~ U[0, 1]`.  Same batch-size adaptation, low-acceptance
    warning and `max_sampling_time` / `return_partial_on_timeout` behaviour; returns (samples, acceptance rate).

    MI355X-first: the potential of all candidates of an iteration is ONE launch of the batched log_prob kernel (one
    x_o broadcast), the ratio test and the compaction of the accepted candidates into a preallocated buffer are
    device ops (stable prefix-sum scatter: same samples in the same order as the reference's boolean-mask gather +
    list + cat), and the loop reads back one integer per iteration."""
    import torch.distributions.transforms as torch_tf

    from sbi_amd.utils.sbiutils import gradient_ascent

    if theta_transform is None:
        theta_transform = torch_tf.IndependentTransform(torch_tf.identity_transform, reinterpreted_batch_ndims=1)
    samples_to_find_max = proposal.sample((num_samples_to_find_max,))

    def potential_over_proposal(theta):
        return potential_fn(theta) - proposal.log_prob(theta)

    _, max_log_ratio = gradient_ascent(
        potential_fn=potential_over_proposal, inits=samples_to_find_max, theta_transform=theta_transform,
        num_iter=num_iter_to_find_max, learning_rate=0.01,
        num_to_optimize=max(1, int(num_samples_to_find_max / 10)), show_progress_bars=False,
    )
    if m < 1.0:
        warnings.warn("A value of m < 1.0 will lead to systematically wrong results.", stacklevel=2)
    log_envelope = max_log_ratio.detach() + torch.log(torch.as_tensor(m))   # proposal.log_prob + this >= potential

    with torch.no_grad():
        out: Optional[Tensor] = None
        num_sampled_total, num_remaining = 0, num_samples
        acceptance_rate = float("nan")
        leakage_warning_raised = False
        sampling_batch_size = min(num_samples, max_sampling_batch_size)
        start_time = time.time()
        filled = 0
        while num_remaining > 0:
            if max_sampling_time is not None and (time.time() - start_time) > max_sampling_time:
                if return_partial_on_timeout and filled > 0:
                    warnings.warn(f"Timeout exceeded after collecting {filled}/{num_samples} samples. Returning "
                                  "partial results.", stacklevel=2)
                    return out[:filled], torch.as_tensor(acceptance_rate)
                raise RuntimeError(
                    "Sampling aborted early because rejection sampling exceeded max_sampling_time. This is likely "
                    "due to extremely low acceptance. You can disable rejection sampling using "
                    "`reject_outside_prior=False` to draw samples directly from the trained estimator. Consider "
                    "switching to MCMC or VI, or checking for model misspecification."
                )
            candidates = proposal.sample((sampling_batch_size,)).reshape(sampling_batch_size, -1)
            log_ratio = potential_fn(candidates) - (proposal.log_prob(candidates) + log_envelope.to(candidates.device))
            target_proposal_ratio = torch.exp(log_ratio).reshape(-1)
            if target_proposal_ratio.is_cuda:
                uniform_rand = torch.rand(target_proposal_ratio.shape, device=target_proposal_ratio.device)
            else:
                uniform_rand = torch.rand(target_proposal_ratio.shape).to(device)
            accept = target_proposal_ratio > uniform_rand
            if out is None:   # one extra row: the dump slot of rejected / surplus candidates
                buf = torch.empty((num_samples + 1, candidates.shape[1]), dtype=candidates.dtype,
                                  device=candidates.device)
                out = buf[:num_samples]
            acc_i = accept.to(torch.long)
            dest = torch.cumsum(acc_i, dim=0) - 1 + filled
            dest = torch.where(accept & (dest < num_samples), dest, torch.full_like(dest, num_samples))
            buf.index_copy_(0, dest, candidates)
            n_acc = int(acc_i.sum().item())                    # the one host read-back of the iteration
            filled = min(filled + n_acc, num_samples)
            num_sampled_total += sampling_batch_size
            num_remaining -= n_acc
            acceptance_rate = (num_samples - num_remaining) / num_sampled_total
            sampling_batch_size = min(max_sampling_batch_size,
                                      max(int(1.5 * num_remaining / max(acceptance_rate, 1e-12)), 100))
            if num_sampled_total > 1000 and acceptance_rate < warn_acceptance and not leakage_warning_raised:
                logging.warning(
                    f"Only {acceptance_rate:.3%} proposal samples were accepted. It may take a long time to collect "
                    f"the remaining {num_remaining} samples. You can prevent long runtimes by setting "
                    "`max_sampling_time` to limit runtime, or disabling rejection sampling (e.g. via "
                    "`reject_outside_prior=False` in `posterior.sample()` when available). Alternatively, consider "
                    "switching to a different sampling method with `build_posterior(..., sample_with='mcmc')`."
                )
                leakage_warning_raised = True
        assert out is not None and filled == num_samples, "Number of accepted samples must match required samples."
    return out, torch.as_tensor(acceptance_rate)
