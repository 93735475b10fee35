"""Flow Matching Posterior Estimation with a device-resident training loop.

API mirror of sbi's ``FMPE`` (sbi/inference/trainers/vfpe/fmpe.py:30-213) on top of ``VectorFieldTrainer.train``
(sbi/inference/trainers/vfpe/base_vf_inference.py:206-350): Adam(5e-4), batch 200, 10 % validation split,
global-norm clip 5.0, validation loss evaluated at fixed times (`validation_times` points between
t_min + nugget and t_max - nugget, every validation batch repeated once per time, :498-536), exponential moving
averages of the epoch losses (:598-636) and the statistical early-stopping rule (:352-407).

As in ``sbi_amd``'s NPE loop the data stay in HBM, a training step is the fused HIP forward+backward +
clip/Adam (``FusedFMPEStep``) and the host reads one pair of scalars per epoch.  Single-round only, like sbi.
"""

from __future__ import annotations

import time
import warnings
from copy import deepcopy
from typing import Callable, Optional, Union

import torch
from torch import Tensor

from sbi_amd.inference.trainers.npe.npe import PosteriorEstimatorTrainer, TrainConfig
from sbi_amd.neural_nets.estimators.flowmatching_estimator import FlowMatchingEstimator
from sbi_amd.utils.collectives import all_reduce_sum


def posterior_flow_nn(model: str = "mlp", z_score_theta: Optional[str] = "independent",
                      z_score_x: Optional[str] = "independent", hidden_features: int = 100, num_layers: int = 5,
                      t_embedding_dim: int = 32, **kwargs) -> Callable:
    """sbi/neural_nets/factory.py:531-620 for the default MLP."""
    if model != "mlp":
        raise NotImplementedError(f"sbi_amd FMPE implements model='mlp' only, got {model!r}")

    def build_fn(batch_theta: Tensor, batch_x: Tensor) -> FlowMatchingEstimator:
        from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

        return build_flow_matching_estimator(batch_theta, batch_x, z_score_theta=z_score_theta, z_score_x=z_score_x,
                                             hidden_features=hidden_features, num_layers=num_layers,
                                             time_embedding_dim=t_embedding_dim, **kwargs)

    return build_fn


class FMPE(PosteriorEstimatorTrainer):
    def __init__(self, prior=None, vf_estimator: Union[str, Callable, None] = None, density_estimator=None,
                 device: str = "cpu", logging_level: Union[int, str] = "WARNING", summary_writer=None, tracker=None,
                 show_progress_bars: bool = True):
        if density_estimator is not None:
            if vf_estimator is not None:
                raise ValueError("Cannot pass both `density_estimator` and `vf_estimator`. Use `vf_estimator` only; "
                                 "`density_estimator` is deprecated.")
            warnings.warn("`density_estimator` is deprecated and will be removed in a future release. Use "
                          "`vf_estimator` instead.", FutureWarning, stacklevel=2)
            vf_estimator = density_estimator
        if vf_estimator is None:
            builder = posterior_flow_nn()
        elif isinstance(vf_estimator, str):
            warnings.warn("Passing a string for `vf_estimator` is deprecated.", FutureWarning, stacklevel=2)
            builder = posterior_flow_nn(model=vf_estimator)
        else:
            builder = vf_estimator
        super().__init__(prior=prior, density_estimator=builder, device=device, logging_level=logging_level,
                         summary_writer=summary_writer, tracker=tracker, show_progress_bars=show_progress_bars)

    def append_simulations(self, theta, x, proposal=None, exclude_invalid_x=None, data_device=None):
        if proposal is not None and proposal is not self._prior:
            raise NotImplementedError("Multi-round FMPE with arbitrary proposals is not implemented")
        return super().append_simulations(theta, x, None, exclude_invalid_x, data_device)

    def train(self, training_batch_size: int = 200, learning_rate: float = 5e-4, validation_fraction: float = 0.1,
              stop_after_epochs: int = 20, max_num_epochs: int = 2**31 - 1, clip_max_norm: Optional[float] = 5.0,
              calibration_kernel: Optional[Callable] = None, ema_loss_decay: float = 0.1,
              validation_times: Union[Tensor, int] = 10, validation_times_nugget: float = 0.05,
              resume_training: bool = False, force_first_round_loss: bool = False,
              discard_prior_samples: bool = False, retrain_from_scratch: bool = False,
              show_train_summary: bool = False, dataloader_kwargs: Optional[dict] = None) -> FlowMatchingEstimator:
        if len(self._data_round_index) == 0:
            raise RuntimeError("No simulations found. You must call .append_simulations() before calling .train().")
        if dataloader_kwargs:
            raise NotImplementedError("The device-resident loop has no DataLoader; dataloader_kwargs is unsupported.")
        cfg = TrainConfig(training_batch_size=training_batch_size, learning_rate=learning_rate,
                          validation_fraction=validation_fraction, stop_after_epochs=stop_after_epochs,
                          max_num_epochs=max_num_epochs, clip_max_norm=clip_max_norm,
                          resume_training=resume_training, retrain_from_scratch=retrain_from_scratch,
                          show_train_summary=show_train_summary)
        theta, x, _ = self.get_simulations(0)
        n = theta.shape[0]
        n_train = int((1 - cfg.validation_fraction) * n)
        n_val = n - n_train
        if not cfg.resume_training or self.train_indices is None:
            perm = self._bcast(torch.randperm(n))
            self.train_indices, self.val_indices = perm[:n_train], perm[n_train:]
        if self._neural_net is None or cfg.retrain_from_scratch:
            self._neural_net = self._build_neural_net(theta[self.train_indices.to(theta.device)].cpu(),
                                                      x[self.train_indices.to(x.device)].cpu())
            if not hasattr(self._neural_net, "loss") or not hasattr(self._neural_net, "solve_schedule"):
                raise TypeError("The vf_estimator builder must return a vector-field estimator (loss, "
                                "solve_schedule, t_min, t_max).")
            self._stepper = None
        net = self._neural_net.to(self._device)
        d = self._dist()
        if d is not None:
            for p in list(net.parameters()) + list(net.buffers()):
                p.data.copy_(self._bcast(p.data))
        theta_d, x_d = theta.to(self._device).float().contiguous(), x.to(self._device).float().contiguous()
        train_idx, val_idx = self.train_indices.to(self._device), self.val_indices.to(self._device)
        rank, world = self._rank_world()
        # sbi_amd's own estimator takes the fused HIP step (it refuses CPU tensors: no fallback); any other
        # vector-field estimator a user supplies (sbi lets `vf_estimator` be a custom builder) is trained
        # through autograd with the same loop, split, all-reduce and clipping
        fused = isinstance(net, FlowMatchingEstimator)
        params = [p for p in net.parameters() if p.requires_grad]
        if not cfg.resume_training or (fused and self._stepper is None) or (not fused and self.optimizer is None):
            if fused:
                from sbi_amd.inference.trainers.fused import FusedFMPEStep

                self._stepper = FusedFMPEStep(net, lr=cfg.learning_rate, clip_max_norm=cfg.clip_max_norm,
                                              distributed=d is not None)
            else:
                self.optimizer = torch.optim.Adam(params, lr=cfg.learning_rate)
            self.epoch, self._val_loss = 0, float("Inf")
        if isinstance(validation_times, int):
            validation_times = net.solve_schedule(validation_times, t_min=net.t_min + validation_times_nugget,
                                                  t_max=net.t_max - validation_times_nugget)
        vtimes = torch.as_tensor(validation_times, dtype=torch.float32, device=self._device)
        T = vtimes.numel()
        B = min(cfg.training_batch_size, n_train)
        Bv = min(cfg.training_batch_size, n_val)
        n_train_batches, n_val_batches = n_train // B, (n_val // Bv if Bv > 0 else 0)
        if n_train_batches == 0 or n_val_batches == 0:
            raise ValueError("Not enough simulations for one training and one validation batch.")

        def my_slice(idx: Tensor) -> Tensor:
            if world == 1:
                return idx
            per = (idx.numel() + world - 1) // world
            return idx[rank * per : min((rank + 1) * per, idx.numel())]

        perm_of = self._epoch_permutations()
        # fused steps on a ROCm device: the epoch's order is never materialised, one launch per minibatch gathers its
        # rows in a fresh keyed pseudo-random order of the training split (utils/shuffle.py, as in NPE.train)
        sampler = None
        if fused and torch.device(self._device).type == "cuda":
            from sbi_amd.utils.shuffle import ShuffledGather

            sg_seed = int(self._bcast(torch.randint(0, 2**62, (1,), dtype=torch.int64)).item())
            sampler = ShuffledGather(theta_d, x_d, train_idx, sg_seed)
        while self.epoch <= cfg.max_num_epochs and not self._converged(self.epoch, cfg.stop_after_epochs):
            t0 = time.time()
            if sampler is None:
                order = perm_of(n_train)
                epoch_idx = train_idx[order]
            sums = torch.zeros(2, device=self._device)
            for b in range(n_train_batches):
                if sampler is not None:
                    from sbi_amd.utils.shuffle import rank_window

                    th, xx = sampler.batch(self.epoch, *rank_window(b * B, B, rank, world))
                else:
                    idx = my_slice(epoch_idx[b * B : (b + 1) * B])
                    th, xx = theta_d.index_select(0, idx), x_d.index_select(0, idx)
                rw = calibration_kernel(xx).float() if calibration_kernel is not None else None
                if fused:
                    losses = self._stepper.loss_and_grad(th, xx, global_batch=B, row_weight=rw)
                    self._stepper.apply()
                else:
                    self.optimizer.zero_grad()
                    losses = net.loss(th, xx)
                    ((losses * rw).sum() / B if rw is not None else losses.sum() / B).backward()
                    if d is not None:
                        for p_ in params:
                            all_reduce_sum(d, p_.grad)
                    if cfg.clip_max_norm is not None:
                        torch.nn.utils.clip_grad_norm_(params, max_norm=cfg.clip_max_norm)
                    self.optimizer.step()
                    losses = losses.detach()
                sums[0] += (losses * rw).sum() if rw is not None else losses.sum()
            val_epoch_idx = val_idx[perm_of(n_val)]
            with torch.no_grad():
                for b in range(n_val_batches):
                    idx = my_slice(val_epoch_idx[b * Bv : (b + 1) * Bv])
                    # every validation row at every validation time (base_vf_inference.py:512-527)
                    th = theta_d.index_select(0, idx).repeat(T, 1)
                    xx = x_d.index_select(0, idx).repeat(T, 1)
                    tt = vtimes.repeat_interleave(idx.numel())
                    losses = net.loss(th, xx, times=tt)
                    if calibration_kernel is not None:
                        losses = losses * calibration_kernel(xx)
                    sums[1] += losses.sum()
            if d is not None:
                all_reduce_sum(d, sums)
            host = sums.cpu()
            if not torch.isfinite(host).all():
                raise AssertionError("NaN/Inf present in FMPE loss.")
            train_loss = float(host[0]) / (n_train_batches * B)
            val_loss = float(host[1]) / (n_val_batches * Bv * T)
            # exponential moving averages (base_vf_inference.py:598-636); convergence looks at the smoothed value
            tl, vl = self._summary["training_loss"], self._summary["validation_loss"]
            train_ema = train_loss if not tl else (1.0 - ema_loss_decay) * tl[-1] + ema_loss_decay * train_loss
            val_ema = val_loss if not vl else (1.0 - ema_loss_decay) * vl[-1] + ema_loss_decay * val_loss
            self._val_loss = val_loss
            tl.append(train_ema)
            vl.append(val_ema)
            self._summary["epoch_durations_sec"].append(time.time() - t0)
            self.epoch += 1
            if self._show_progress_bars and rank == 0:
                print("\r", f"Training neural network. Epochs trained: {self.epoch}", end="")
        if self.epoch > cfg.max_num_epochs:
            if self._val_loss < self._best_val_loss:
                self._best_val_loss = self._val_loss
                self._best_model_state_dict = deepcopy(net.state_dict())
            elif self._best_model_state_dict is not None:
                self._load_state(net, self._best_model_state_dict)
            warnings.warn("Maximum number of epochs `max_num_epochs={}` reached, but network has not yet fully "
                          "converged. Consider increasing it.".format(cfg.max_num_epochs), stacklevel=2)
        elif self._show_progress_bars and rank == 0:
            print(f"\n Neural network successfully converged after {self.epoch} epochs.")
        self._summary["epochs_trained"].append(self.epoch)
        self._summary["best_validation_loss"].append(self._best_val_loss)
        if rank == 0:
            self._summarize(0)      # FMPE is single-round (fmpe.py:128-145)
        return deepcopy(net)

    def _converged(self, epoch: int, stop_after_epochs: int) -> bool:
        """base_vf_inference.py:352-407: an epoch only counts as fruitless when the validation loss is more than
        two running standard deviations above the best one."""
        net = self._neural_net
        if epoch == 0:
            self._best_val_loss = float("inf")
            self._epochs_since_last_improvement = 0
            self._best_model_state_dict = None
        if self._val_loss < self._best_val_loss:
            self._best_val_loss = self._val_loss
            self._epochs_since_last_improvement = 0
            self._best_model_state_dict = deepcopy(net.state_dict())
        else:
            hist = self._summary["validation_loss"]
            if len(hist) < stop_after_epochs:
                return False
            std = torch.tensor(hist[-stop_after_epochs * 2 :]).std().item()
            if (self._val_loss - self._best_val_loss) / std > 2.0:
                self._epochs_since_last_improvement += 1
            else:
                self._epochs_since_last_improvement = 0
        if self._epochs_since_last_improvement > stop_after_epochs - 1:
            if self._best_model_state_dict is not None:
                self._load_state(net, self._best_model_state_dict)
            return True
        return False

    def build_posterior(self, vector_field_estimator: Optional[FlowMatchingEstimator] = None, prior=None,
                        sample_with: str = "ode", **kwargs):
        from sbi_amd.inference.posteriors.vector_field_posterior import VectorFieldPosterior

        if sample_with != "ode":
            raise NotImplementedError("sbi_amd FMPE posterior samples with the probability-flow ODE only")
        est = vector_field_estimator if vector_field_estimator is not None else self._neural_net
        if est is None:
            raise ValueError("train() first or pass a vector_field_estimator")
        prior = prior if prior is not None else self._prior
        self._posterior = VectorFieldPosterior(deepcopy(est).to(self._device), prior, device=str(self._device))
        return self._posterior
