"""``MCMCPosterior`` -- sample a potential with the device-resident vectorised slice sampler.

Mirror of sbi/inference/posteriors/mcmc_posterior.py:41-368, :517-812 for ``method="slice_np_vectorized"``
(sbi's default; ``"slice_np"`` runs the same sampler here -- the serial numpy variant has no reason to exist
on a GPU).  Sampling happens in the unconstrained / z-scored space of ``theta_transform``
(``mcmc_transform(prior)``), chains are initialised by ``proposal`` / ``sir`` / ``resample`` /
``latest_sample`` for all chains at once, every tick evaluates the potential of all chains with the batched
log_prob kernel (single x_o broadcast, 44 B per evaluation) and advances them with
``sbi_amd_mcmc_slice_tick``.  Pyro / PyMC samplers are not part of this path.
"""

from __future__ import annotations

from math import ceil
from typing import Any, Callable, Dict, Optional, Union

import torch
from torch import Tensor

from sbi_amd.neural_nets.estimators.shape_handling import reshape_to_batch_event
from sbi_amd.samplers.mcmc import SliceSamplerVectorized, proposal_init, resample_given_potential_fn, sir_init
from sbi_amd.utils.sbiutils import mcmc_transform
from sbi_amd.utils.torchutils import ensure_theta_batched, process_device

_SLICE_METHODS = ("slice_np", "slice_np_vectorized")
_OTHER_METHODS = ("hmc_pyro", "nuts_pyro", "slice_pymc", "hmc_pymc", "nuts_pymc")


def unconstrained_potential(potential_fn, transform, device):
    """The slice sampler walks UNCONSTRAINED space: returns u -> potential of the constrained point T^-1(u), corrected
    by the volume change of the map (what mcmc_posterior.py:953-986 gets from utils/potentialutils.py:15-51)."""

    def density_of(u) -> Tensor:
        u = ensure_theta_batched(torch.as_tensor(u, dtype=torch.float32)).to(device)
        constrained = transform.inv(u)
        volume = transform.log_abs_det_jacobian(constrained, u).to(device)
        return potential_fn(constrained, track_gradients=False).to(device) - volume

    return density_of


def _process_thin_default(thin: int) -> int:
    """mcmc_posterior.py:1069-1082: the default changed from 10 to 1 upstream; -1 selects it."""
    return 1 if thin == -1 else thin


class MCMCPosterior:
    def __init__(self, potential_fn: Callable, proposal: Any, theta_transform=None,
                 method: str = "slice_np_vectorized", thin: int = -1, warmup_steps: int = 200, num_chains: int = 20,
                 init_strategy: str = "resample", init_strategy_parameters: Optional[Dict[str, Any]] = None,
                 num_workers: int = 1, mp_context: str = "spawn", device: Optional[str] = None,
                 x_shape: Optional[torch.Size] = None):
        if method not in _SLICE_METHODS + _OTHER_METHODS:
            raise NameError(f"The sampling method {method} is not implemented!")
        self.potential_fn = potential_fn
        self.proposal = proposal
        if device is None:
            device = getattr(potential_fn, "device", "cpu")
        self._device = process_device(device)
        # keep the constrained <- unconstrained direction and build its inverse on demand: an
        # `_InverseTransform` is tied to its parent through a weak reference that deepcopy / pickling breaks
        tt = torch.distributions.transforms.identity_transform if theta_transform is None else theta_transform
        self._to_constrained = tt.inv
        self.method = method
        self.thin = _process_thin_default(thin)
        self.warmup_steps = warmup_steps
        self.num_chains = num_chains
        self.init_strategy = init_strategy
        self.init_strategy_parameters = init_strategy_parameters or {}
        self.num_workers = num_workers
        self.mp_context = mp_context
        self._posterior_sampler = None
        self._mcmc_init_params: Optional[Tensor] = None
        self._x: Optional[Tensor] = None
        self._x_shape = x_shape
        self._purpose = "It provides MCMC to .sample() from the posterior and can evaluate the _unnormalized_ " \
                        "posterior density with .log_prob()."

    @property
    def theta_transform(self):
        """constrained -> unconstrained (what `mcmc_transform` returns)."""
        # `.inv` of a plain transform builds its `_InverseTransform`; `.inv` of an `_InverseTransform` hands back the
        # parent it wraps -- either way the transform the caller passed in (wrapping unconditionally double-inverted a
        # plain transform and raised NotImplementedError in .sample())
        return self._to_constrained.inv

    # -- x handling (base_posterior.py:170-214) ------------------------------------------
    @property
    def default_x(self) -> Optional[Tensor]:
        return self._x

    def set_default_x(self, x: Tensor) -> "MCMCPosterior":
        x = torch.as_tensor(x, dtype=torch.float32)
        if not torch.isfinite(x).all():
            raise ValueError("x_o contains NaN or Inf values.")
        self._x = x.to(self._device)
        return self

    def _x_else_default_x(self, x: Optional[Tensor]) -> Tensor:
        if x is not None:
            return torch.as_tensor(x, dtype=torch.float32).to(self._device)
        if self._x is None:
            raise ValueError("Context `x` needed when a default has not been set. If you'd like to have a default, "
                             "use the `.set_default_x()` method.")
        return self._x

    @property
    def mcmc_method(self) -> str:
        return self.method

    def set_mcmc_method(self, method: str) -> "MCMCPosterior":
        self.method = method
        return self

    @property
    def posterior_sampler(self):
        return self._posterior_sampler

    # -- density --------------------------------------------------------------------------
    def log_prob(self, theta: Tensor, x: Optional[Tensor] = None, track_gradients: bool = False) -> Tensor:
        """The potential: the UNNORMALISED posterior log-density (mcmc_posterior.py:208-235)."""
        import warnings

        warnings.warn("`.log_prob()` is deprecated for methods that can only evaluate the log-probability up to a "
                      "normalizing constant. Use `.potential()` instead.", stacklevel=2)
        return self.potential(theta, x, track_gradients)

    def potential(self, theta: Tensor, x: Optional[Tensor] = None, track_gradients: bool = False) -> Tensor:
        self.potential_fn.set_x(self._x_else_default_x(x))
        theta = ensure_theta_batched(torch.as_tensor(theta)).to(self._device)
        return self.potential_fn(theta, track_gradients=track_gradients)

    # -- sampling -------------------------------------------------------------------------
    def sample(self, sample_shape=torch.Size(), x: Optional[Tensor] = None, method: Optional[str] = None,
               thin: Optional[int] = None, warmup_steps: Optional[int] = None, num_chains: Optional[int] = None,
               init_strategy: Optional[str] = None, init_strategy_parameters: Optional[Dict[str, Any]] = None,
               num_workers: Optional[int] = None, mp_context: Optional[str] = None,
               show_progress_bars: bool = True, **kwargs) -> Tensor:
        x = self._x_else_default_x(x)
        self.potential_fn.set_x(x, x_is_iid=True)
        method = self.method if method is None else method
        thin = self.thin if thin is None else _process_thin_default(thin)
        warmup_steps = self.warmup_steps if warmup_steps is None else warmup_steps
        num_chains = self.num_chains if num_chains is None else num_chains
        init_strategy = self.init_strategy if init_strategy is None else init_strategy
        init_strategy_parameters = (self.init_strategy_parameters if init_strategy_parameters is None
                                    else init_strategy_parameters)
        if method in _OTHER_METHODS:
            raise NotImplementedError(f"method={method!r}: the Pyro / PyMC samplers are outside the accelerated "
                                      "path; use 'slice_np_vectorized'.")
        if method not in _SLICE_METHODS:
            raise NameError(f"The sampling method {method} is not implemented!")

        potential_ = unconstrained_potential(self.potential_fn, self.theta_transform, self._device)

        fused = self._fused_potential()
        if fused is not None:
            potential_ = fused
        self.potential_ = potential_
        initial_params = self._get_initial_params(init_strategy, num_chains, **init_strategy_parameters)
        num_samples = torch.Size(sample_shape).numel()
        with torch.no_grad():
            transformed = self._slice_np_mcmc(num_samples, potential_, initial_params, thin, warmup_steps)
        samples = self.theta_transform.inv(transformed)
        return samples.reshape((*torch.Size(sample_shape), -1))

    def __getstate__(self):
        """`potential_` is the closure the last `sample()` call ran its chains on (the reference keeps a picklable
        `partial` there, mcmc_posterior.py:145); it is rebuilt by every `sample()` call, so it is simply not part of
        the pickled state.  The live object is left untouched (tests/save_and_load_test.py:23-45)."""
        state = dict(self.__dict__)
        state["potential_"] = None
        state["_posterior_sampler"] = None      # holds the same closure as its log_prob_fn (diagnostics only)
        return state

    def _fused_potential(self) -> Optional[Callable]:
        """Four launches per tick instead of ~15: when the potential is the NSF estimator's log-prob inside the
        prior support and the parameter transform is one `mcmc_transform` builds (z-scoring of an unbounded
        prior, logit map of a box, identity on an unbounded prior), theta = T^-1(u) and log|det| come from
        `sbi_amd_mcmc_to_constrained`, log q from the batched log_prob kernel, and the subtraction happens inside
        the tick kernel.  Returns None when anything does not match (the generic path is always correct)."""
        import torch.distributions.transforms as tf
        from torch.distributions import constraints

        from sbi_amd import _lib
        from sbi_amd.inference.potentials.posterior_based_potential import PosteriorBasedPotential
        from sbi_amd.neural_nets.estimators.nsf_flow import NSFFlow

        pot = self.potential_fn
        if not isinstance(pot, PosteriorBasedPotential) or not isinstance(pot.posterior_estimator, NSFFlow):
            return None
        if torch.device(self._device).type != "cuda":
            return None
        x_o = reshape_to_batch_event(pot.x_o, pot.posterior_estimator.condition_shape)
        if x_o.shape[0] != 1:
            return None
        try:
            support = pot.prior.support
        except (NotImplementedError, AttributeError):
            return None
        base_c = support.base_constraint if hasattr(support, "base_constraint") else support
        t = self._to_constrained                         # unconstrained -> constrained
        if isinstance(t, tf.IndependentTransform):
            t = t.base_transform
        D = pot.posterior_estimator.input_shape[0]
        dev = torch.device(self._device)

        def vec(v):
            return torch.as_tensor(v, dtype=torch.float32, device=dev).expand(D).contiguous()

        unbounded = isinstance(base_c, constraints._Real)
        if isinstance(t, tf.ComposeTransform) and len(t.parts) == 0 and unbounded:
            kind, p0, p1 = 0, None, None
        elif isinstance(t, tf.AffineTransform) and unbounded:
            kind, p0, p1 = 1, vec(t.loc), vec(t.scale)
        elif (isinstance(t, tf.ComposeTransform) and len(t.parts) == 2 and isinstance(t.parts[0], tf.SigmoidTransform)
              and isinstance(t.parts[1], tf.AffineTransform) and isinstance(base_c, constraints._Interval)):
            low, high = vec(base_c.lower_bound), vec(base_c.upper_bound)
            p0, p1 = vec(t.parts[1].loc), vec(t.parts[1].scale)
            if not (torch.allclose(p0, low) and torch.allclose(p0 + p1, high)):
                return None                              # the box of the transform is not the prior's support
            kind = 2
        else:
            return None
        lib = _lib.load()
        net = pot.posterior_estimator.net
        # the kernels take the EMBEDDED condition (standardizing_net -> embedding_net in front of the flow,
        # flow.py:1395-1416): embed x_o once, outside the chain loop -- never hand raw x to the kernel
        est = pot.posterior_estimator
        with torch.no_grad():
            x_row = est._embed(x_o.to(dev)).reshape(1, -1).to(torch.float32).contiguous()
        if x_row.shape[1] != net.hyper.C:
            return None

        def potential_(u: Tensor):
            u = u.to(torch.float32).contiguous()
            C = u.shape[0]
            theta = torch.empty_like(u)
            lad = torch.empty(C, dtype=torch.float32, device=u.device)
            with torch.cuda.device(u.device):
                rc = lib.sbi_amd_mcmc_to_constrained(kind, C, D, _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(u),
                                                     _lib.ptr(theta), _lib.ptr(lad), _lib.current_stream(u.device))
            _lib.check(rc, "mcmc_to_constrained")
            logp, _ = est._kernel_log_prob(theta, x_row, False)
            return logp, lad

        def log_q(theta: Tensor) -> Tensor:      # the estimator's log-density at constrained points, one launch
            return est._kernel_log_prob(theta, x_row, False)[0]

        # the slice sampler's tick kernel applies the transform itself (sbi_amd_mcmc_slice_tick): two launches per tick
        potential_.fused_spec = (kind, p0, p1, log_q, net, x_row)
        return potential_

    def _get_initial_params(self, init_strategy: str, num_chains: int, **kwargs) -> Tensor:
        """mcmc_posterior.py:517-659, all chains in one batched call."""
        if init_strategy == "proposal":
            init = proposal_init(self.proposal, transform=self.theta_transform, num_chains=num_chains, **kwargs)
        elif init_strategy == "sir":
            init = sir_init(self.proposal, self.potential_fn, transform=self.theta_transform, num_chains=num_chains,
                            **kwargs)
        elif init_strategy == "resample":
            init = resample_given_potential_fn(self.proposal, self.potential_fn, transform=self.theta_transform,
                                               num_chains=num_chains, **kwargs)
        elif init_strategy == "latest_sample":
            stored = self._mcmc_init_params
            if stored is None:
                raise ValueError("`init_strategy='latest_sample'` continues the chains of an earlier `sample()` call, "
                                 "but this posterior holds no chain states. Use another init strategy, for example "
                                 "'proposal' or 'sir'.")
            if num_chains > stored.shape[0]:
                raise ValueError(f"`init_strategy='latest_sample'` has {stored.shape[0]} chain state(s) from the last "
                                 f"run, but this call needs {num_chains}. Run at most {stored.shape[0]} chain(s), or "
                                 "use another init strategy.")
            init = stored[:num_chains]
        else:
            raise NotImplementedError(f"Init strategy {init_strategy} is not implemented.")
        init = init.reshape(num_chains, -1).to(self._device)
        assert init.shape[0] == num_chains, "Initial params shape mismatch."
        return init

    def _slice_np_mcmc(self, num_samples: int, potential_function: Callable, initial_params: Tensor, thin: int,
                       warmup_steps: int, init_width: float = 0.01) -> Tensor:
        """mcmc_posterior.py:737-811."""
        num_chains, dim_samples = initial_params.shape
        sampler = SliceSamplerVectorized(init_params=initial_params, log_prob_fn=potential_function,
                                         num_chains=num_chains, thin=thin, verbose=False, init_width=init_width)
        warmup_ = warmup_steps * thin
        num_samples_ = ceil((num_samples * thin) / num_chains)
        samples = sampler.run(warmup_ + num_samples_)            # chains x samples x dim (already thinned)
        samples = samples[:, warmup_steps:, :]                   # discard warmup steps
        self._posterior_sampler = sampler
        self._mcmc_init_params = samples[:, -1, :].reshape(num_chains, dim_samples)
        samples = samples.reshape(-1, dim_samples)[:num_samples]  # chains are interchangeable
        return samples.to(torch.float32).to(self._device)
