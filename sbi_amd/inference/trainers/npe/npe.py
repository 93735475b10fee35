"""Single-round Neural Posterior Estimation with a device-resident training loop.

API mirror of sbi's ``NPE`` (= ``NPE_C``) for the first-round / maximum-likelihood
case: ``NPE(prior, density_estimator, device).append_simulations(theta, x).train(...)``
then ``build_posterior()``  (sbi/inference/trainers/npe/npe_base.py:81-163, 188-299,
301-418, 420-511; loop semantics of sbi/inference/trainers/base.py:499-563,
1060-1284: Adam(lr 5e-4), batch 200, 10 % validation split, global-norm clip 5.0,
early stopping after 20 non-improving epochs, best-weights restore, drop_last).

MI355X-first: theta/x live in HBM, minibatches are device-side index gathers over a
per-epoch ``randperm`` (no DataLoader, no per-row Python), every step is the fused
HIP forward+backward + clip/Adam (``FusedTrainStep``), losses accumulate on the
device and the host reads ONE pair of scalars per epoch (the reference syncs every
minibatch, base.py:1179,1218).  Under ``torch.distributed`` every rank takes a
contiguous slice of each global batch and gradients are summed with one flat
all-reduce over RCCL.
"""

from __future__ import annotations

import logging
import time
import warnings
from copy import deepcopy
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import torch
from torch import Tensor, nn
from torch.distributions import Distribution

from sbi_amd.neural_nets.estimators.base import ConditionalDensityEstimator
from sbi_amd.neural_nets.estimators.nsf_flow import NSFFlow
from sbi_amd.neural_nets.factory import posterior_nn
from sbi_amd.neural_nets.net_builders.estimator_configs import NSFConfig, ZukoNSFConfig
from sbi_amd.utils.collectives import all_reduce_sum
from sbi_amd.utils.sbiutils import handle_invalid_x, warn_on_invalid_x
from sbi_amd.utils.torchutils import check_if_prior_on_device, process_device


@dataclass(frozen=True)
class TrainConfig:
    """Validated training hyper-parameters (sbi/inference/trainers/_contracts.py:48-92)."""

    training_batch_size: int = 200
    learning_rate: float = 5e-4
    validation_fraction: float = 0.1
    stop_after_epochs: int = 20
    max_num_epochs: int = 2**31 - 1
    clip_max_norm: Optional[float] = 5.0
    resume_training: bool = False
    retrain_from_scratch: bool = False
    show_train_summary: bool = False

    def __post_init__(self):
        if self.training_batch_size < 1:
            raise ValueError("training_batch_size must be a positive integer")
        if not (0.0 < self.validation_fraction < 1.0):
            raise ValueError("validation_fraction must be in (0, 1)")
        if self.learning_rate <= 0:
            raise ValueError("learning_rate must be positive")
        if self.stop_after_epochs < 1 or self.max_num_epochs < 0:
            raise ValueError("stop_after_epochs must be >= 1 and max_num_epochs >= 0")
        if self.clip_max_norm is not None and self.clip_max_norm <= 0:
            raise ValueError("clip_max_norm must be positive or None")


def check_estimator_arg(estimator) -> None:
    """Same rejections as sbi/utils/user_input_checks.py:678-705."""
    if isinstance(estimator, nn.Module):
        raise TypeError(
            "The `density_estimator` must be a string, a config object or a function that builds the network "
            "from (theta, x) -- not an already-built nn.Module."
        )
    if isinstance(estimator, type):
        raise TypeError("Pass a config *instance* (e.g. NSFConfig()), not the config class.")
    if not (isinstance(estimator, (str, NSFConfig, ZukoNSFConfig)) or callable(estimator)):
        raise TypeError(f"Unsupported density_estimator argument of type {type(estimator).__name__}")


def validate_theta_and_x(theta: Any, x: Any, data_device: str = "cpu", training_device: str = "cpu"):
    """fp32 tensors with equal batch size, moved to the data device (user_input_checks.py:708-764)."""
    if not isinstance(theta, Tensor) or not isinstance(x, Tensor):
        raise AssertionError("Parameters theta and simulation outputs x must be `torch.Tensor`s.")
    if theta.shape[0] != x.shape[0]:
        raise AssertionError(f"Number of parameter sets (={theta.shape[0]}) must match the number of simulation "
                             f"outputs (={x.shape[0]})")
    if theta.dtype != torch.float32 or x.dtype != torch.float32:
        raise AssertionError("Type of parameters and simulation outputs must be float32.")
    if str(data_device) != str(training_device) and data_device != "cpu":
        warnings.warn(f"Data on '{data_device}' but training on '{training_device}'.", stacklevel=2)
    return theta.to(data_device), x.to(data_device)


class ImproperEmpirical:
    """Stand-in prior when none was given: support = everything (sbiutils.py:1012-1117)."""

    def __init__(self, values: Tensor):
        self._values = values

    def sample(self, sample_shape=torch.Size()):
        n = torch.Size(sample_shape).numel()
        idx = torch.randint(0, self._values.shape[0], (n,), device=self._values.device)
        return self._values[idx].reshape(*sample_shape, *self._values.shape[1:])

    def log_prob(self, value: Tensor) -> Tensor:
        return torch.zeros(value.shape[:-1], device=value.device)


class PosteriorEstimatorTrainer:
    def __init__(self, prior: Optional[Distribution] = None,
                 density_estimator: Union[str, NSFConfig, Callable, None] = None, device: str = "cpu",
                 logging_level: Union[int, str] = "WARNING", summary_writer=None, tracker=None,
                 show_progress_bars: bool = True):
        self._device = process_device(device)
        check_if_prior_on_device(self._device, prior)
        self._prior = prior
        self._show_progress_bars = show_progress_bars
        self._tracker = tracker
        logging.getLogger().setLevel(logging_level if isinstance(logging_level, int) else logging_level.upper())
        if density_estimator is not None:
            check_estimator_arg(density_estimator)
        if density_estimator is None:
            # The reference defaults to MAF; this package implements the NSF path only.
            self._build_neural_net = NSFConfig().build
        elif isinstance(density_estimator, str):
            warnings.warn(
                "Passing a string for `density_estimator` is deprecated. Use a per-model config instead, e.g. "
                "`from sbi_amd.neural_nets import NSFConfig`.", FutureWarning, stacklevel=3,
            )
            self._build_neural_net = posterior_nn(model=density_estimator)
        elif isinstance(density_estimator, (NSFConfig, ZukoNSFConfig)):
            self._build_neural_net = density_estimator.build
        else:
            self._build_neural_net = density_estimator

        self._neural_net: Optional[ConditionalDensityEstimator] = None
        self._theta_roundwise: List[Tensor] = []
        self._x_roundwise: List[Tensor] = []
        self._prior_masks: List[Tensor] = []
        self._data_round_index: List[int] = []
        self._proposal_roundwise: List[Any] = []
        self._round = 0
        self._val_loss = float("Inf")
        self._best_val_loss = float("Inf")
        self._epochs_since_last_improvement = 0
        self._best_model_state_dict = None
        self.epoch = 0
        self.optimizer = None
        self._stepper = None
        self.train_indices: Optional[Tensor] = None
        self.val_indices: Optional[Tensor] = None
        self._summary: Dict[str, list] = dict(epochs_trained=[], best_validation_loss=[], validation_loss=[],
                                              training_loss=[], epoch_durations_sec=[])

    # ------------------------------------------------------------------ data
    def append_simulations(self, theta: Tensor, x: Tensor, proposal=None, exclude_invalid_x: Optional[bool] = None,
                           data_device: Optional[str] = None) -> "PosteriorEstimatorTrainer":
        # round bookkeeping (npe_base.py:222-240): data sampled from the prior is round 0 (MLE loss); any
        # other proposal opens a new round trained with the proposal-corrected atomic loss
        restricted_prior = (type(proposal).__name__ == "RestrictedPrior"
                            and getattr(proposal, "_prior", None) is self._prior)   # npe_base.py:223-229
        if proposal is None or proposal is self._prior or restricted_prior:
            current_round = 0
        elif not self._data_round_index:
            current_round = 1
        else:
            current_round = max(self._data_round_index) + 1
        if exclude_invalid_x is None:
            exclude_invalid_x = current_round == 0
        if data_device is None:
            data_device = self._device
        theta, x = validate_theta_and_x(theta, x, data_device=data_device, training_device=self._device)
        is_valid_x, num_nans, num_infs = handle_invalid_x(x, exclude_invalid_x=exclude_invalid_x)
        if current_round > 0 and (num_nans + num_infs) > 0:
            # the atomic loss normalises across the batch (sbiutils.py:553-578): invalid rows can neither stay
            # nor be dropped silently
            algorithm = f"Multiround {type(self).__name__}"
            if not exclude_invalid_x:
                raise ValueError(
                    f"Found {num_nans} NaN simulations and {num_infs} Inf simulations. {algorithm} does not allow "
                    "invalid simulations. Replace the invalid values with an unreasonably low or high value."
                )
            logging.warning(
                f"Found {num_nans} NaN simulations and {num_infs} Inf simulations. These will be discarded from "
                f"training due to `exclude_invalid_x=True`. Please be aware that this gives systematically wrong "
                f"results for {algorithm} and is only recommended for expert users."
            )
        else:
            warn_on_invalid_x(num_nans, num_infs, exclude_invalid_x)
        x, theta = x[is_valid_x], theta[is_valid_x]
        self._check_proposal(proposal)
        self._data_round_index.append(current_round)
        self._theta_roundwise.append(theta)
        self._x_roundwise.append(x)
        # True where theta came from the prior (sbiutils.py:602-613)
        self._prior_masks.append(torch.full((theta.shape[0], 1), current_round == 0, dtype=torch.bool))
        self._proposal_roundwise.append(proposal)
        if self._prior is None or isinstance(self._prior, ImproperEmpirical):
            if proposal is not None:
                raise ValueError(
                    "You did not pass a prior at initialization, but now you passed a proposal. If you want to run "
                    "multi-round NPE, you have to specify a prior (set the `.prior` argument or re-initialize the "
                    "object with a prior distribution). If the samples you passed to `append_simulations()` were "
                    "sampled from the prior, you can run single-round inference with "
                    "`append_simulations(..., proposal=None)`."
                )
            self._prior = ImproperEmpirical(self.get_simulations()[0].to(self._device))
        return self

    def _check_proposal(self, proposal) -> None:
        """npe_base.py:577-610."""
        if proposal is None:
            return
        if hasattr(proposal, "default_x") and proposal.default_x is None:
            raise ValueError(
                "`proposal.default_x` is None, i.e. there is no x_o for training. Set it with "
                "`posterior.set_default_x(x_o)`."
            )
        if (hasattr(proposal, "posterior_estimator") and self._neural_net is not None
                and proposal.posterior_estimator is self._neural_net):
            raise ValueError(
                "The proposal's posterior_estimator is the same object as the trainer's neural network. This will "
                "cause incorrect training because the proposal's weights will change during optimization. Use "
                "`deepcopy(estimator)` when creating the proposal, or use `trainer.build_posterior()` which "
                "handles this automatically."
            )
        is_neural = hasattr(proposal, "posterior_estimator") or hasattr(
            getattr(proposal, "potential_fn", None), "posterior_estimator")   # MCMC / rejection posteriors
        if not is_neural:
            warnings.warn(
                "The proposal you passed is neither the prior nor a neural posterior: the atomic multi-round loss "
                "will be used. If the parameters were sampled from the prior, pass proposal=None.", stacklevel=3,
            )

    def get_simulations(self, starting_round: int = 0):
        th = torch.cat([t for t, r in zip(self._theta_roundwise, self._data_round_index) if r >= starting_round])
        xx = torch.cat([t for t, r in zip(self._x_roundwise, self._data_round_index) if r >= starting_round])
        mk = torch.cat([t for t, r in zip(self._prior_masks, self._data_round_index) if r >= starting_round])
        return th, xx, mk

    # ------------------------------------------------------------------ distributed helpers
    @staticmethod
    def _dist():
        import torch.distributed as dist

        return dist if (dist.is_available() and dist.is_initialized()) else None

    def _rank_world(self):
        d = self._dist()
        return (d.get_rank(), d.get_world_size()) if d is not None else (0, 1)

    def _bcast(self, t: Tensor) -> Tensor:
        d = self._dist()
        if d is not None:
            from sbi_amd.utils.collectives import device_capable

            buf = t.to(self._device)
            backend_dev = self._device if (buf.device.type == "cuda" and device_capable(d)) else "cpu"
            buf = t.to(backend_dev)
            d.broadcast(buf, src=0)
            t = buf.to(t.device)
        return t

    def _epoch_permutations(self):
        """Returns `perm(n) -> device LongTensor`: the per-epoch SubsetRandomSampler orders.  On a ROCm device the
        permutation is drawn ON the device from a generator seeded once per train() call from torch's global
        RNG on rank 0 (seed broadcast: every rank draws identical permutations with no per-epoch collective and no
        host work -- a CPU `torch.randperm(90000)` alone costs more than a 65 536-row training step).  On the CPU
        (gloo tests) it stays the global-RNG host permutation."""
        if torch.device(self._device).type != "cuda":
            return lambda n: self._bcast(torch.randperm(n)).to(self._device)
        seed = int(self._bcast(torch.randint(0, 2**62, (1,), dtype=torch.int64)).item())
        gen = torch.Generator(device=self._device)
        gen.manual_seed(seed)
        # permutations are drawn EIGHT AT A TIME per length: argsort of 31-bit random keys (four radix passes; the
        # handful of ties among 1e5 keys fall back to index order), one batched sort -- a `torch.randperm` is a dozen
        # launches, two per epoch was a tenth of the host's enqueue time of an epoch
        stock: Dict[int, list] = {}

        def perm(n: int) -> Tensor:
            have = stock.setdefault(n, [])
            if not have:
                keys = torch.randint(0, 2**31 - 1, (8, n), generator=gen, device=self._device, dtype=torch.int32)
                have.extend(keys.argsort(dim=1).unbind(0))
            return have.pop()

        return perm

    # ------------------------------------------------------------------ training
    def train(self, num_atoms: int = 10, training_batch_size: int = 200, learning_rate: float = 5e-4,
              validation_fraction: float = 0.1, stop_after_epochs: int = 20, max_num_epochs: int = 2**31 - 1,
              clip_max_norm: Optional[float] = 5.0, calibration_kernel: Optional[Callable] = None,
              resume_training: bool = False, force_first_round_loss: bool = False,
              discard_prior_samples: bool = False, use_combined_loss: bool = False,
              retrain_from_scratch: bool = False, show_train_summary: bool = False,
              dataloader_kwargs: Optional[dict] = None) -> ConditionalDensityEstimator:
        if len(self._data_round_index) == 0:
            raise RuntimeError("No simulations found. You must call .append_simulations() before calling .train().")
        if dataloader_kwargs:
            raise NotImplementedError("The device-resident loop has no DataLoader; dataloader_kwargs is unsupported.")
        # npe_c.py:194-223 / npe_base.py:629-662
        self._num_atoms = num_atoms
        self._use_combined_loss = use_combined_loss
        self._round = max(self._data_round_index)
        if self._round == 0 and self._neural_net is not None and not (force_first_round_loss or resume_training):
            raise ValueError(
                "This neural network has already been trained. If you want to continue training without adding new "
                "simulations, set resume_training=True. If you appended new simulations, you must either provide a "
                "proposal in append_simulations(), or set force_first_round_loss=True for simulations drawn from "
                "the prior. Warning: Setting force_first_round_loss=True with simulations not drawn from the prior "
                "will produce the proposal posterior instead of the true posterior, which is typically more narrow."
            )
        atomic = self._round > 0 and not force_first_round_loss
        if atomic:
            print(f"Using {type(self).__name__} with atomic loss")
        start_idx = int(discard_prior_samples and self._round > 0)
        cfg = TrainConfig(training_batch_size=training_batch_size, learning_rate=learning_rate,
                          validation_fraction=validation_fraction, stop_after_epochs=stop_after_epochs,
                          max_num_epochs=max_num_epochs, clip_max_norm=clip_max_norm,
                          resume_training=resume_training, retrain_from_scratch=retrain_from_scratch,
                          show_train_summary=show_train_summary)
        theta, x, prior_masks = self.get_simulations(start_idx)
        n = theta.shape[0]
        n_train = int((1 - cfg.validation_fraction) * n)
        n_val = n - n_train
        if not cfg.resume_training or self.train_indices is None:
            perm = self._bcast(torch.randperm(n))       # identical split on every rank
            self.train_indices, self.val_indices = perm[:n_train], perm[n_train:]

        # build the network from the TRAINING split on the CPU (npe_base.py:674-708)
        if self._neural_net is None or cfg.retrain_from_scratch:
            self._neural_net = self._build_neural_net(theta[self.train_indices.to(theta.device)].cpu(),
                                                      x[self.train_indices.to(x.device)].cpu())
            if not isinstance(self._neural_net, nn.Module) or not hasattr(self._neural_net, "loss"):
                raise TypeError("The density_estimator builder must return a ConditionalDensityEstimator.")
            self._stepper = None
        net = self._neural_net.to(self._device)
        params = [p for p in net.parameters() if p.requires_grad]
        if not params:
            raise TypeError(f"{type(self).__name__} cannot train {type(net).__name__}: it has no trainable "
                            "parameters.")
        d = self._dist()
        if d is not None:   # replicas start identical
            for p in list(net.parameters()) + list(net.buffers()):
                p.data.copy_(self._bcast(p.data))

        theta_d = theta.to(self._device)
        x_d = x.to(self._device)
        masks_d = prior_masks.to(self._device)
        prior = self._prior
        train_idx = self.train_indices.to(self._device)
        val_idx = self.val_indices.to(self._device)
        rank, world = self._rank_world()

        # the fused optimizer step owns ONE flat parameter buffer: an embedding net with trainable weights takes the
        # autograd path (bridge kernels incl. d loss / d embedded x + torch Adam over all parameters)
        # (a frozen / parameter-free embedding still has to be APPLIED: FusedTrainStep embeds under no_grad)
        emb_trainable = any(p.requires_grad for p in net.embedding_net.parameters()) \
            if getattr(net, "embedding_net", None) is not None else False
        fused = (isinstance(net, NSFFlow) and torch.device(self._device).type == "cuda" and calibration_kernel is None
                 and not emb_trainable and (not atomic or getattr(net.net, "supports_atomic", False)))
        if not cfg.resume_training or (fused and self._stepper is None) or (not fused and self.optimizer is None):
            if fused:
                from sbi_amd.inference.trainers.fused import FusedTrainStep

                self._stepper = FusedTrainStep(net, lr=cfg.learning_rate, clip_max_norm=cfg.clip_max_norm,
                                               distributed=d is not None)
            else:
                self.optimizer = torch.optim.Adam(params, lr=cfg.learning_rate)
            self.epoch, self._val_loss = 0, float("Inf")

        B = min(cfg.training_batch_size, n_train)
        Bv = min(cfg.training_batch_size, n_val)
        n_train_batches = n_train // B
        n_val_batches = n_val // Bv if Bv > 0 else 0
        if n_train_batches == 0 or n_val_batches == 0:
            raise ValueError("Not enough simulations for one training and one validation batch.")

        def my_slice(idx: Tensor) -> Tensor:   # this rank's contiguous share of a global batch
            if world == 1:
                return idx
            per = (idx.numel() + world - 1) // world
            return idx[rank * per : min((rank + 1) * per, idx.numel())]

        from sbi_amd.inference.trainers.npe.atomic import log_prob_proposal_posterior_atomic

        def net_losses(th: Tensor, xx: Tensor, mk: Tensor) -> Tensor:
            """npe_base.py:542-575: MLE in the first round, proposal-corrected atomic loss afterwards."""
            if not atomic:
                return net.loss(th, xx)
            return -log_prob_proposal_posterior_atomic(net, prior, th, xx, mk, self._num_atoms,
                                                       self._use_combined_loss)

        def batch_losses(idx: Tensor, train: bool, global_batch: int) -> Tensor:
            th, xx = theta_d.index_select(0, idx), x_d.index_select(0, idx)
            mk = masks_d.index_select(0, idx) if atomic else None      # (only the atomic loss reads the prior masks)
            if fused:
                if train:
                    if atomic:
                        return self._stepper.atomic_step(th, xx, mk, prior, self._num_atoms, self._use_combined_loss,
                                                         global_batch=global_batch)
                    return self._stepper.step(th, xx, global_batch=global_batch)
                with torch.no_grad():
                    return net_losses(th, xx, mk)
            if train:
                self.optimizer.zero_grad()
                losses = net_losses(th, xx, mk)
                if not torch.isfinite(losses).all():
                    raise AssertionError("NaN/Inf present in NPE loss.")
                if calibration_kernel is not None:
                    losses = losses * calibration_kernel(xx)
                (losses.sum() / global_batch).backward()
                if d is not None:
                    for p in params:
                        all_reduce_sum(d, p.grad)
                if cfg.clip_max_norm is not None:
                    torch.nn.utils.clip_grad_norm_(params, max_norm=cfg.clip_max_norm)
                self.optimizer.step()
                return losses.detach()
            with torch.no_grad():
                losses = net_losses(th, xx, mk)
                return losses * calibration_kernel(xx) if calibration_kernel is not None else losses

        perm_of = self._epoch_permutations()
        import os as _os

        # Fused MLE steps on a ROCm device: the epoch's order is never materialised -- one launch per minibatch
        # gathers the batch's rows in a fresh keyed pseudo-random order of the training split (utils/shuffle.py;
        # SubsetRandomSampler + collation of base.py:541-560).  The atomic loss also needs the prior masks of the batch:
        # it takes the batch's source rows from the same sampler (`indices`) and gathers with them; the autograd path
        # keeps the argsort permutations.
        sampler = None
        # (structured x in front of a frozen / parameter-free embedding net -- (N, c, h, w) images through nn.Flatten or
        # a frozen CNN -- keeps the index path: the gather kernel moves flat fp32 rows)
        if fused and x_d.dim() == 2 and theta_d.dim() == 2 and x_d.dtype == theta_d.dtype == torch.float32:
            from sbi_amd.utils.shuffle import ShuffledGather

            sg_seed = int(self._bcast(torch.randint(0, 2**62, (1,), dtype=torch.int64)).item())
            sampler = ShuffledGather(theta_d, x_d, train_idx, sg_seed)

        def my_range(lo: int, count: int):     # this rank's contiguous share of rows lo .. lo + count of an order
            from sbi_amd.utils.shuffle import rank_window

            return rank_window(lo, count, rank, world)

        host_ring = []       # filled below when `pipelined` (pinned float32 pairs: train / validation loss sums)
        pipelined = fused and _os.environ.get("SBI_AMD_EAGER_EPOCH_SYNC") != "1"      # (why: see the loop below)
        from sbi_amd import _lib as _lib_mod

        # Per-epoch bookkeeping on the device, kept to a handful of launches whatever the number of batches: the per-row
        # losses of every batch are collected and summed ONCE per epoch (cat + sum per split, instead of a sum and an add
        # per batch); when the validation batches tile the split its rows are gathered once per train() call; snapshots
        # go into a ring of preallocated buffers (no allocation per epoch).
        sums_ring = [torch.zeros(2, device=self._device) for _ in range(4)]      # (an epoch's pair is read back one epoch late)

        def loss_sums(train_rows: list, val_rows: list, out: Tensor) -> Tensor:
            for i, parts in enumerate((train_rows, val_rows)):
                torch.sum(parts[0] if len(parts) == 1 else torch.cat(parts), dim=0, keepdim=True, out=out[i : i + 1])
            return out

        # (`val_fixed` is a second device copy of the validation split -- n_val (D + C) floats, 800 KB at 10 000 rows --
        #  alive for the duration of this train() call: it saves one gather launch per validation batch and epoch)
        val_fixed = None
        if fused and n_val_batches * Bv == n_val and world == 1 and not atomic:      # (fused: no calibration kernel)
            val_fixed = (theta_d.index_select(0, val_idx), x_d.index_select(0, val_idx))

        def val_batch_losses(b: int, val_epoch_idx) -> Tensor:
            if val_fixed is not None:
                with torch.no_grad():
                    return net_losses(val_fixed[0][b * Bv : (b + 1) * Bv], val_fixed[1][b * Bv : (b + 1) * Bv], None)
            return batch_losses(my_slice(val_epoch_idx[b * Bv : (b + 1) * Bv]), False, Bv)

        snap_ring = [None] * 4       # (at most three epoch records are alive at a time)
        if (fused and not atomic and n_train_batches <= 2 and isinstance(net, NSFFlow) and hasattr(net.net, "hyper")
                and _os.environ.get("SBI_AMD_TAIL_EXTRA", "1") != "0"):
            # one or two steps per epoch, validation batches on the OTHER kernel family (e.g. batch 65 536 / 10 000
            # validation rows): let the step's table pack refresh the validation image too instead of a separate pack
            # launch every epoch
            try:
                lib_ = _lib_mod.load()
                cfg_ = net.net.hyper.c_config()
                # (the kernel family is chosen by the rows ONE rank passes to a call: its share of the global batch)
                rows_tr, rows_va = max(my_range(0, B)[1], 1), max(my_range(0, Bv)[1], 1)
                k_tr, k_va = lib_.sbi_amd_nsf_image_kind(cfg_, int(rows_tr), 1), lib_.sbi_amd_nsf_image_kind(cfg_, int(rows_va), 0)
                self._stepper.tail_extra_images = (2 if k_va == 1 else 1) if (k_tr >= 0 and k_va >= 0 and k_tr != k_va) else 0
            except (AttributeError, RuntimeError):
                self._stepper.tail_extra_images = 0
        elif fused:
            self._stepper.tail_extra_images = 0

        def launch_epoch(e: int) -> dict:
            """Enqueue one epoch's device work (training steps, validation pass, [loss all-reduce]); nothing here
            waits for the device.  Returns the record `finish_epoch` turns into the epoch's host bookkeeping."""
            rec = {"epoch": e, "t0": time.time()}
            if pipelined:      # the epoch's own device time (the host clock would also count the NEXT epoch's enqueue)
                rec["ev0"] = torch.cuda.Event(enable_timing=True)
                rec["ev0"].record()
            net.train()
            tr_rows, va_rows = [], []
            if sampler is not None and not atomic:
                for b in range(n_train_batches):
                    th, xx = sampler.batch(e, *my_range(b * B, B))
                    tr_rows.append(self._stepper.step(th, xx, global_batch=B))
            elif sampler is not None:
                for b in range(n_train_batches):
                    tr_rows.append(batch_losses(sampler.indices(e, *my_range(b * B, B)), True, B))
            else:
                order = perm_of(n_train)   # SubsetRandomSampler
                epoch_idx = train_idx[order]
                for b in range(n_train_batches):
                    idx = my_slice(epoch_idx[b * B : (b + 1) * B])
                    tr_rows.append(batch_losses(idx, True, B))
            if pipelined:      # weights + optimizer state after this epoch's steps
                rec["snap"] = snap_ring[e % len(snap_ring)] = self._stepper.snapshot_into(snap_ring[e % len(snap_ring)])
            net.eval()
            # (every validation row is used when the batches tile the split exactly: the order of a sum is immaterial)
            val_epoch_idx = val_idx if n_val_batches * Bv == n_val else val_idx[perm_of(n_val)]
            for b in range(n_val_batches):
                va_rows.append(val_batch_losses(b, val_epoch_idx))
            sums = loss_sums(tr_rows, va_rows, sums_ring[e % len(sums_ring)])
            if d is not None:
                all_reduce_sum(d, sums)
            if pipelined:
                # one of four pinned read-back buffers allocated once per train() call: at most three epoch records
                # are alive at a time (finishing, in flight, speculative).  A fresh `torch.empty(pin_memory=True)` per
                # epoch goes to hipHostMalloc whenever the pinned-block cache has no block whose last use has provably
                # finished -- a third of a millisecond, on whichever epochs lose that race
                rec["host"] = host_ring[e % len(host_ring)]
                rec["host"].copy_(sums, non_blocking=True)
                rec["event"] = torch.cuda.Event(enable_timing=True)
                rec["event"].record()
            else:
                rec["host"] = sums.cpu()                        # the one sync of the epoch
            return rec

        def finish_epoch(rec: dict) -> None:
            if "event" in rec:
                # Poll instead of `synchronize()`: the blocking wait goes to sleep on the completion signal and, on some
                # hosts of the pool, wakes up several hundred microseconds after it fired -- with one epoch of ~1 ms in
                # flight behind this one that is enough for the device to run dry (measured through bench.py's npe_train
                # leg: 1.35 - 1.38 ms per epoch on two boxes, 1.00 on two others, with IDENTICAL per-launch enqueue
                # times).  The loop spins for at most the remainder of one epoch.  SBI_AMD_EVENT_SPIN=0: the blocking wait.
                # The spin is BOUNDED (it holds the GIL and a core): a few multiples of the previous epoch's device time,
                # at least 2 ms and at most 50 ms, then the blocking wait takes over -- long epochs (where the wake-up
                # latency is noise) and a hung device end up sleeping, not spinning.  SBI_AMD_EVENT_SPIN=0: never spin.
                if _os.environ.get("SBI_AMD_EVENT_SPIN", "1") != "0":
                    durs = self._summary["epoch_durations_sec"]
                    budget = min(0.05, max(0.002, 4.0 * durs[-1])) if durs else 0.002
                    t_spin = time.perf_counter()
                    while not rec["event"].query():
                        if time.perf_counter() - t_spin > budget:
                            rec["event"].synchronize()
                            break
                else:
                    rec["event"].synchronize()
            host = rec["host"]
            if not torch.isfinite(host).all():
                if "event" in rec and last_good["snap"] is not None:
                    # pipelined loop: the losses are read one epoch late, so the live weights have already been
                    # stepped on this epoch's (and possibly the speculative epoch's) non-finite gradients: put the
                    # weights and optimizer state of the last finite epoch back before giving up
                    torch.cuda.synchronize(self._device)
                    self._stepper.restore(last_good["snap"])
                raise AssertionError("NaN/Inf present in NPE loss.")
            if "snap" in rec:
                last_good["snap"] = rec["snap"]
            train_loss = float(host[0]) / (n_train_batches * B)
            self._val_loss = float(host[1]) / (n_val_batches * Bv)
            self._summary["training_loss"].append(train_loss)
            self._summary["validation_loss"].append(self._val_loss)
            self._summary["epoch_durations_sec"].append(
                rec["ev0"].elapsed_time(rec["event"]) * 1e-3 if "ev0" in rec else time.time() - rec["t0"])
            self.epoch = rec["epoch"] + 1
            if self._show_progress_bars and rank == 0:
                print("\r", f"Training neural network. Epochs trained: {self.epoch}", end="")

        # Fused path: epochs are software-pipelined against the host.  Epoch e+1 is enqueued BEFORE the losses of
        # epoch e are read back, so the device never waits for the early-stopping bookkeeping (one pinned-memory
        # read per epoch, one epoch late).  The decisions are the reference's, in the reference's order: the
        # weights + optimizer state after every epoch are snapshotted on the device, `_converged` scores epoch e with
        # that snapshot, and if it says stop, the speculative epoch e+1 is thrown away (its summary entries are never
        # written, the optimizer state is rolled back, the best weights restored as always).
        # Where this differs from the eager loop (SBI_AMD_EAGER_EPOCH_SYNC=1) and from the reference's loop:
        #  * a speculative epoch is always enqueued, so whatever it draws (the contrastive indices of the atomic loss,
        #    the key of its sampler order) is consumed even when the epoch is discarded: torch's RNG state after
        #    train() is not the eager loop's;
        #  * the NaN / Inf check of epoch e fires after epoch e + 1 was enqueued: the weights and the optimizer state
        #    are put back to the last epoch whose losses were finite before the AssertionError is raised;
        #  * `epoch_durations_sec` are device times between events around an epoch's own launches (the host clock
        #    would include the next epoch's enqueue).
        if pipelined:
            host_ring.extend(torch.empty(2, dtype=torch.float32, pin_memory=True) for _ in range(4))
        last_good = {"snap": None}
        if not pipelined:
            while self.epoch <= cfg.max_num_epochs and not self._converged(self.epoch, cfg.stop_after_epochs):
                finish_epoch(launch_epoch(self.epoch))
        else:
            in_flight = None
            while True:
                if in_flight is None:
                    if self.epoch > cfg.max_num_epochs or self._converged(self.epoch, cfg.stop_after_epochs):
                        break
                    in_flight = launch_epoch(self.epoch)
                    continue
                nxt = in_flight["epoch"] + 1
                spec = launch_epoch(nxt) if nxt <= cfg.max_num_epochs else None
                finish_epoch(in_flight)                      # -> self.epoch == nxt, self._val_loss of that epoch
                if self.epoch > cfg.max_num_epochs:
                    break
                if self._converged(self.epoch, cfg.stop_after_epochs, snapshot=in_flight["snap"]):
                    if spec is not None:
                        spec["event"].synchronize()
                        self._stepper.restore_optimizer(in_flight["snap"])
                    break
                in_flight = spec

        if self.epoch > cfg.max_num_epochs:
            # the final epoch was never scored by `_converged` (base.py:1122-1129)
            if self._val_loss < self._best_val_loss:
                self._best_val_loss = self._val_loss
                self._best_model_state_dict = deepcopy(PosteriorEstimatorTrainer._native_state(net))
            elif self._best_model_state_dict is not None:
                self._load_state(net, self._best_model_state_dict)
            warnings.warn("Maximum number of epochs `max_num_epochs={}` reached, but network has not yet fully "
                          "converged. Consider increasing it.".format(cfg.max_num_epochs), stacklevel=2)
        elif self._show_progress_bars and rank == 0:
            print(f"\n Neural network successfully converged after {self.epoch} epochs.")
        self._summary["epochs_trained"].append(self.epoch)
        self._summary["best_validation_loss"].append(self._best_val_loss)
        if rank == 0:
            self._summarize(self._round)
        if cfg.show_train_summary and rank == 0:
            print(self._describe_round())
        net.zero_grad(set_to_none=True)
        return deepcopy(net)

    @staticmethod
    def _native_state(net: nn.Module):
        """The estimator's state in the kernels' own two-tensor form (the public `state_dict()` speaks nflows' key
        names: ~100 per-layer copies, not something to build once per improved epoch)."""
        inner = getattr(net, "net", None)
        if inner is None or not hasattr(inner, "native_state_dict"):
            return net.state_dict()
        inner._native_state_dict = True
        try:
            return net.state_dict()
        finally:
            inner._native_state_dict = False

    @staticmethod
    def _load_state(net: nn.Module, sd) -> None:
        net.load_state_dict(sd)
        inner = getattr(net, "net", None)
        if inner is not None:
            inner.__dict__.pop("_packed_cache", None)

    def _converged(self, epoch: int, stop_after_epochs: int, snapshot: Optional[dict] = None) -> bool:
        """Early stopping with best-weights bookkeeping (base.py:1254-1284).  `snapshot` (pipelined fused loop): the
        device copy of the flat parameters the scored epoch ended with -- the live ones may already belong to the
        next, speculative epoch."""
        converged = False
        net = self._neural_net
        if epoch == 0 or self._val_loss < self._best_val_loss:
            self._best_val_loss = self._val_loss
            self._epochs_since_last_improvement = 0
            if snapshot is None:
                self._best_model_state_dict = deepcopy(PosteriorEstimatorTrainer._native_state(net))
            else:
                native = PosteriorEstimatorTrainer._native_state(net)
                self._best_model_state_dict = type(native)(
                    (k, snapshot["params"].clone() if k == "net.flat_params" else v.clone())
                    for k, v in native.items())
        else:
            self._epochs_since_last_improvement += 1
        if self._epochs_since_last_improvement > stop_after_epochs - 1:
            self._load_state(net, self._best_model_state_dict)
            converged = True
        return converged

    def _summarize(self, round_: int) -> None:
        """Write the round's statistics to the experiment tracker the caller passed as `tracker=` (protocol
        sbi/sbi_types.py:73-91; calls, tags and step numbering of trainers/base.py:1317-1385).  Without one nothing is
        written: the reference's default -- a TensorBoard directory under sbi-logs/ -- belongs to its logging
        subsystem, which is outside this path."""
        t = self._tracker
        if t is None:
            return
        sm = self._summary
        t.log_metric(name="epochs_trained", value=sm["epochs_trained"][-1], step=round_ + 1)
        t.log_metric(name="best_validation_loss", value=sm["best_validation_loss"][-1], step=round_ + 1)
        offset = int(sum(sm["epochs_trained"][:-1]))
        for tag in ("validation_loss", "training_loss", "epoch_durations_sec"):
            for i, v in enumerate(sm[tag][offset:]):
                t.log_metric(name=tag, value=v, step=int(offset + i))
        t.flush()

    def _describe_round(self) -> str:
        s = self._summary
        return (f"\n -------------------------\n ||||| ROUND 1 STATS |||||:\n -------------------------\n"
                f" Epochs trained: {s['epochs_trained'][-1]}\n"
                f" Best validation performance: {-s['best_validation_loss'][-1]:.4f}\n"
                " -------------------------\n")

    @property
    def summary(self):
        return self._summary

    # ------------------------------------------------------------------ posterior
    def build_posterior(self, density_estimator: Optional[ConditionalDensityEstimator] = None,
                        prior: Optional[Distribution] = None, sample_with: str = "direct",
                        direct_sampling_parameters: Optional[Dict[str, Any]] = None, **kwargs):
        from sbi_amd.inference.posteriors.direct_posterior import DirectPosterior

        if sample_with not in ("direct", "mcmc", "rejection"):
            raise NotImplementedError(
                f"sample_with={sample_with!r}: VI / importance posteriors are outside the accelerated path "
                "(SURVEY.md section 8f-3 covers the batched-log_prob callers: 'direct', 'mcmc', 'rejection')."
            )
        if prior is None:
            if self._prior is None:
                raise ValueError("You did not pass a prior. You have to pass the prior either at initialization "
                                 "`inference = NPE(prior)` or to `.build_posterior(prior=prior)`.")
            prior = self._prior
        else:
            check_if_prior_on_device(self._device, prior)
        if density_estimator is None:
            if self._neural_net is None:
                raise ValueError("No trained estimator: call .train() first or pass density_estimator=...")
            estimator = deepcopy(self._neural_net)
            device = self._device
        else:
            estimator = density_estimator
            device = str(next(density_estimator.parameters()).device)
        if sample_with == "mcmc":
            # npe_base.py:420-512 + posteriors/posterior_parameters.py: potential = estimator log-prob inside the
            # prior support, proposal = prior, sampling in the unconstrained space of mcmc_transform(prior)
            from sbi_amd.inference.posteriors.mcmc_posterior import MCMCPosterior
            from sbi_amd.inference.potentials.posterior_based_potential import posterior_estimator_based_potential
            from sbi_amd.utils.sbiutils import mcmc_transform

            potential_fn, _ = posterior_estimator_based_potential(estimator, prior, x_o=None)
            mcmc_parameters = dict(kwargs.get("mcmc_parameters") or {})
            enable_transform = mcmc_parameters.pop("enable_transform", True)
            theta_transform = mcmc_transform(prior, device=device, enable_transform=enable_transform)
            self._posterior = MCMCPosterior(potential_fn=potential_fn, proposal=prior, theta_transform=theta_transform,
                                            method=kwargs.get("mcmc_method", "slice_np_vectorized"), device=device,
                                            **mcmc_parameters)
            return deepcopy(self._posterior)
        if sample_with == "rejection":
            # trainers/base.py:794-796, 1029-1036: potential = estimator log-prob inside the prior support,
            # proposal = prior, RejectionPosteriorParameters(max_sampling_batch_size, num_samples_to_find_max,
            # num_iter_to_find_max, m)
            from sbi_amd.inference.posteriors.rejection_posterior import RejectionPosterior
            from sbi_amd.inference.potentials.posterior_based_potential import posterior_estimator_based_potential
            from sbi_amd.utils.sbiutils import mcmc_transform

            params = dict(kwargs.get("rejection_sampling_parameters") or {})
            unknown = set(params) - {"max_sampling_batch_size", "num_samples_to_find_max", "num_iter_to_find_max", "m"}
            if unknown:
                raise TypeError(f"unexpected rejection_sampling_parameters: {sorted(unknown)}")
            potential_fn, _ = posterior_estimator_based_potential(estimator, prior, x_o=None)
            self._posterior = RejectionPosterior(potential_fn=potential_fn, proposal=prior,
                                                 theta_transform=mcmc_transform(prior, device=device), device=device,
                                                 **params)
            return deepcopy(self._posterior)
        self._posterior = DirectPosterior(posterior_estimator=estimator, prior=prior, device=device,
                                          **(direct_sampling_parameters or {}))
        return deepcopy(self._posterior)


class NPE_C(PosteriorEstimatorTrainer):
    """NPE-C / APT (npe_c.py:91-440): maximum likelihood in the first round, the atomic proposal-posterior loss
    in later rounds (``append_simulations(theta, x, proposal=posterior)``).  The closed-form MoG correction
    (`_log_prob_proposal_posterior_mog`) applies to mixture-density estimators only and is not part of the
    NSF path."""


NPE = NPE_C   # sbi/inference/__init__.py:22
SNPE = NPE_C
