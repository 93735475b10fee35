"""Test-only adapter: the CPU oracle behind sbi_amd's ConditionalDensityEstimator
contract, so host logic (trainer loop, posterior, sampler, data-parallel plumbing) can be
exercised without a GPU.  Never imported by the product."""

import torch

from oracle.nsf_oracle import NSFOracle
from sbi_amd.neural_nets.estimators.base import ConditionalDensityEstimator


class OracleEstimator(ConditionalDensityEstimator):
    def __init__(self, theta, x, **kw):
        o = NSFOracle(theta, x, **kw)
        super().__init__(o.net, input_shape=theta[0].shape, condition_shape=x[0].shape)
        self._o = [o]   # keep the oracle without registering its parameters twice

    def log_prob(self, input, condition, **kwargs):
        self._check_input_shape(input)
        self._check_condition_shape(condition)
        input, condition, _ = self._broadcast_and_align(input, condition)
        S, B = input.shape[0], input.shape[1]
        lp = self.net.log_prob(input.reshape(S * B, -1), condition.reshape(S * B, -1))
        return lp.reshape(S, B)

    def loss(self, input, condition, **kwargs):
        return -self.log_prob(input.unsqueeze(0), condition)[0]

    def sample(self, sample_shape, condition, **kwargs):
        n = torch.Size(sample_shape).numel()
        s = self.net.sample(n, condition).transpose(0, 1)
        return s.reshape((*sample_shape, condition.shape[0], *self.input_shape))


def oracle_build_fn(**kw):
    def build(theta, x):
        return OracleEstimator(theta, x, **kw)

    return build


class OracleVectorField(torch.nn.Module):
    """The FMPE oracle behind the vector-field-estimator surface the FMPE trainer uses (loss, solve_schedule,
    t_min, t_max).  Times and noise are deterministic functions of the row, so a data-parallel run sees the same
    draws as a single process."""

    def __init__(self, theta, x, **kw):
        super().__init__()
        from oracle.fmpe_oracle import FMPEOracle

        self.o = FMPEOracle(theta.shape[1], x.shape[1], **kw)
        with torch.no_grad():
            self.o.mean_0.copy_(theta.mean(0))
            self.o.std_0.copy_(theta.std(0))
            self.o.x_mean.copy_(x.mean(0))
            self.o.x_std.copy_(x.std(0))
        self.input_shape, self.condition_shape = theta[0].shape, x[0].shape
        self.t_min, self.t_max = 0.0, 1.0

    def solve_schedule(self, steps, t_min=None, t_max=None):
        return torch.linspace(self.t_max if t_max is None else t_max, self.t_min if t_min is None else t_min, steps)

    def loss(self, input, condition, times=None, **kwargs):
        key = input.sum(-1, keepdim=True)
        if times is None:
            times = torch.frac(key[:, 0].abs() * 7.31)
        noise = torch.sin(key * torch.arange(1, input.shape[1] + 1) * 3.7) * 1.3
        return self.o.loss(input, condition, times, noise)


def oracle_vf_build_fn(**kw):
    def build(theta, x):
        return OracleVectorField(theta, x, **kw)

    return build
