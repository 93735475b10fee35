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
