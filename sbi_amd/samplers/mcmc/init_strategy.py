"""Initial positions of MCMC chains (sbi/samplers/mcmc/init_strategy.py:29-114), for ALL chains at once:
the reference calls a one-sample init function `num_chains` times (each weighting 10 000 candidates); here
the `num_chains x num_candidate_samples` candidates go through the batched log_prob kernel in one call."""

from __future__ import annotations

from typing import Any, Callable

import torch
from torch import Tensor


def proposal_init(proposal: Any, transform, num_chains: int = 1, **kwargs: Any) -> Tensor:
    """`num_chains` draws from the proposal, transformed."""
    return transform(proposal.sample((num_chains,)).detach())


@torch.no_grad()
def sir_init(proposal: Any, potential_fn: Callable, transform, num_chains: int = 1,
             num_candidate_samples: int = 10_000, **kwargs: Any) -> Tensor:
    """Sampling importance resampling (Rubin 1988), one winner per chain: weights potential - proposal.log_prob
    (sbi/samplers/importance/sir.py:13-71)."""
    cand = proposal.sample((num_chains * num_candidate_samples,)).detach()
    log_w = potential_fn(cand).detach() - proposal.log_prob(cand).detach()
    return transform(_pick(cand, log_w, num_chains, num_candidate_samples))


@torch.no_grad()
def resample_given_potential_fn(proposal: Any, potential_fn: Callable, transform, num_chains: int = 1,
                                num_candidate_samples: int = 10_000, num_batches: int = 1, **kwargs: Any) -> Tensor:
    """Like SIR but weighted by the potential alone (init_strategy.py:67-114)."""
    n = num_candidate_samples * num_batches
    cand = proposal.sample((num_chains * n,)).detach()
    return transform(_pick(cand, potential_fn(cand).detach(), num_chains, n))


def _pick(cand: Tensor, log_w: Tensor, num_chains: int, per_chain: int) -> Tensor:
    log_w = log_w.reshape(num_chains, per_chain)
    log_w = log_w - torch.logsumexp(log_w, dim=1, keepdim=True)
    probs = torch.exp(log_w)
    probs[~torch.isfinite(probs)] = 0.0
    dead = probs.sum(dim=1, keepdim=True) <= 0            # no candidate inside the support: fall back to uniform
    probs = torch.where(dead, torch.ones_like(probs), probs)
    idx = torch.multinomial(probs, 1).reshape(-1)
    cand = cand.reshape(num_chains, per_chain, -1)
    return cand[torch.arange(num_chains, device=cand.device), idx]
