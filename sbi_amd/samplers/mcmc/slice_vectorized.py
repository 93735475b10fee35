"""Vectorised slice sampler with every chain advanced on the device.

Mirror of ``SliceSamplerVectorized`` (sbi/samplers/mcmc/slice_numpy.py:353-587).  The reference keeps one
Python dict per chain and walks all of them between two batched log-prob evaluations (numpy <-> torch copies
every tick); here the chain states are device tensors and one tick is: evaluate ``log_prob_fn(next_param)``
(the batched log_prob kernel behind the potential) -> ``sbi_amd_mcmc_slice_tick`` (all transitions, one
thread per chain).  The host only looks at the finished-chain counter every ``poll_every`` ticks.
"""

from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import Tensor

from sbi_amd import _lib


class SliceSamplerVectorized:
    def __init__(self, log_prob_fn: Callable[[Tensor], Tensor], init_params: Tensor, num_chains: int = 1,
                 thin: int = 1, tuning: int = 50, verbose: bool = False, init_width: float = 0.01,
                 max_width: float = float("inf"), num_workers: int = 1, poll_every: int = 64):
        self._log_prob_fn = log_prob_fn
        self.x = torch.as_tensor(init_params, dtype=torch.float32).contiguous()
        _lib.require_device(self.x)
        if self.x.dim() != 2 or self.x.shape[0] != num_chains:
            raise ValueError(f"init_params must have shape (num_chains, dim); got {tuple(self.x.shape)}")
        self.num_chains = int(num_chains)
        self.thin = 1 if thin is None else int(thin)
        self.tuning = int(tuning)
        self.verbose = verbose
        self.init_width = float(init_width)
        self.max_width = float(max_width)
        self.poll_every = int(poll_every)
        self.n_dims = self.x.shape[1]
        self._samples: Optional[Tensor] = None
        self.num_ticks = 0

    @torch.no_grad()
    def run(self, num_samples: int) -> Tensor:
        """(num_chains, ceil(num_samples / thin), dim) samples; `tuning` extra sweeps adapt the brackets first."""
        assert num_samples >= 0
        lib = _lib.load()
        dev = self.x.device
        C, D = self.num_chains, self.n_dims
        x = self.x.clone().contiguous()
        nxt = x.clone()
        width = torch.full((C, D), self.init_width, dtype=torch.float32, device=dev)
        order = torch.rand(C, D, device=dev).argsort(dim=1).to(torch.int32).contiguous()
        istate = torch.zeros(C, 4, dtype=torch.int32, device=dev)
        fstate = torch.zeros(C, 8, dtype=torch.float32, device=dev)
        samples = torch.empty(C, max(int(num_samples), 1), D, dtype=torch.float32, device=dev)
        done = torch.zeros(1, dtype=torch.int32, device=dev)
        max_width = self.max_width if self.max_width != float("inf") else 3.0e38
        tick = 0
        fused = getattr(self._log_prob_fn, "fused_spec", None)
        if fused is not None:
            # Two launches per tick: the batched log_prob kernel on the constrained point, then the tick kernel, which
            # also draws its uniforms (Philox keyed by a seed from torch's generator) and maps the NEXT evaluation point
            # to constrained space.  (The reference's sampler draws from NumPy's global generator,
            # slice_numpy.py:353-587: a distribution to match, not a stream.)
            kind, p0, p1, log_q, net, x_row = fused
            seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
            theta = torch.empty_like(nxt)
            lad = torch.empty(C, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.sbi_amd_mcmc_to_constrained(kind, C, D, _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(nxt), _lib.ptr(theta),
                                                     _lib.ptr(lad), _lib.current_stream(dev))
            _lib.check(rc, "mcmc_to_constrained")
            # Persistent form first: `poll_every` ticks of every chain per launch (a workgroup owns 16 chains and
            # alternates their log-density with their tick; sbi_amd_mcmc_slice_run).  Configurations the cooperative
            # kernels do not take fall back to two launches per tick.
            persistent = bool(getattr(self, "persistent", True))
            if persistent:
                from sbi_amd.neural_nets.estimators.nsf_flow import packed_weights

                packed = packed_weights(net, rows=None)
                cfg = net.hyper.c_config()
                scratch = torch.empty(C, dtype=torch.float32, device=dev)
            while persistent:
                with torch.cuda.device(dev):
                    rc = lib.sbi_amd_mcmc_slice_run(cfg, _lib.ptr(packed), _lib.ptr(net.zstats), _lib.ptr(x_row), C,
                                                    int(num_samples), self.tuning, max_width, _lib.ptr(x), _lib.ptr(nxt),
                                                    _lib.ptr(width), _lib.ptr(order), _lib.ptr(istate), _lib.ptr(fstate),
                                                    _lib.ptr(samples), _lib.ptr(done), seed, tick, self.poll_every, kind,
                                                    _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(theta), _lib.ptr(lad),
                                                    _lib.ptr(scratch), _lib.current_stream(dev))
                if rc == _lib.E_UNSUPPORTED and tick == 0:
                    persistent = False
                    break
                _lib.check(rc, "mcmc_slice_run")
                tick += self.poll_every
                if int(done.item()) == C:
                    break
            while not persistent:
                logp = log_q(theta)
                with torch.cuda.device(dev):
                    rc = lib.sbi_amd_mcmc_slice_tick(C, D, int(num_samples), self.tuning, max_width, _lib.ptr(logp),
                                                     _lib.ptr(lad), None, _lib.ptr(x), _lib.ptr(nxt), _lib.ptr(width),
                                                     _lib.ptr(order), _lib.ptr(istate), _lib.ptr(fstate),
                                                     _lib.ptr(samples), _lib.ptr(done), seed, tick, kind, _lib.ptr(p0),
                                                     _lib.ptr(p1), _lib.ptr(theta), _lib.ptr(lad), _lib.current_stream(dev))
                _lib.check(rc, "mcmc_slice_tick")
                tick += 1
                if tick % self.poll_every == 0 and int(done.item()) == C:
                    break
        while fused is None:
            out = self._log_prob_fn(nxt)
            # a (log_prob, offset) pair keeps the potential's "- log|det|" out of a separate launch
            logp, offset = out if isinstance(out, tuple) else (out, None)
            logp = logp.reshape(-1).to(torch.float32).contiguous()
            if logp.numel() != C:
                raise ValueError(f"log_prob_fn returned {logp.numel()} values for {C} chains")
            u = torch.rand(C, 4 + D, device=dev)
            with torch.cuda.device(dev):
                rc = lib.sbi_amd_mcmc_slice_tick(C, D, int(num_samples), self.tuning, max_width, _lib.ptr(logp),
                                                 _lib.ptr(offset), _lib.ptr(u), _lib.ptr(x), _lib.ptr(nxt),
                                                 _lib.ptr(width),
                                                 _lib.ptr(order), _lib.ptr(istate), _lib.ptr(fstate),
                                                 _lib.ptr(samples), _lib.ptr(done), 0, 0, 0, None, None, None, None,
                                                 _lib.current_stream(dev))
            _lib.check(rc, "mcmc_slice_tick")
            tick += 1
            if tick % self.poll_every == 0 and int(done.item()) == C:
                break
        self.num_ticks = tick
        self.x = x
        self.width = width
        samples = samples[:, : int(num_samples)][:, :: self.thin, :]
        self._samples = samples
        return samples

    def get_samples(self, num_samples: Optional[int] = None, group_by_chain: bool = True) -> Tensor:
        if self._samples is None:
            raise ValueError("No samples found from MCMC run.")
        if group_by_chain:
            return self._samples if num_samples is None else self._samples[:, -num_samples:, :]
        flat = self._samples.reshape(-1, self._samples.shape[-1])
        return flat if num_samples is None else flat[-num_samples:]
