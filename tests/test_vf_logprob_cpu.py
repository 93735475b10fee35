"""Host logic of VectorFieldPosterior.log_prob on the CPU (no kernels): a stand-in estimator with an ANALYTIC vector field
v(theta, t; x) = -a(x) * theta (divergence -a D, flow theta_1 = theta_0 e^{-a}) through the same code path -- augmented
state layout, direction of integration, sign of the log-det term, base density, prior-support masking, iid observations,
chunking, argument refusals -- and the host-side Dormand-Prince stepper the CPU path uses."""
import math

import pytest
import torch
from torch.distributions import Independent, Normal, Uniform

from sbi_amd.inference.posteriors.vector_field_posterior import VectorFieldPosterior


class LinearField:
    """Quacks like FlowMatchingEstimator for the posterior: theta' = -a theta with a = 0.3 + x.sum()."""

    input_shape = torch.Size([3])
    condition_shape = torch.Size([2])
    t_min, t_max = 0.0, 1.0

    def __init__(self):
        self.mean_base = torch.zeros(1, 3)
        self.std_base = torch.ones(1, 3)
        self.calls = 0

    def rate(self, x):
        return 0.3 + float(x.reshape(-1, 2)[0].sum())

    def ode_fn(self, theta, x, t):
        return -self.rate(x) * theta

    def ode_fn_and_divergence(self, theta, x, t, v_out=None, div_out=None):
        self.calls += 1
        a = self.rate(x)
        v_out.copy_(-a * theta)
        div_out.fill_(-a * theta.shape[1])
        return v_out, div_out


def exact(theta, a):
    z = theta * math.exp(-a)
    return (-0.5 * z * z - 0.5 * math.log(2 * math.pi)).sum(-1) - a * theta.shape[1]


def test_log_prob_matches_the_analytic_flow_and_masks_the_prior_support():
    est = LinearField()
    prior = Independent(Uniform(-2.0 * torch.ones(3), 2.0 * torch.ones(3)), 1)
    post = VectorFieldPosterior(est, prior, device="cpu")
    x = torch.tensor([[0.2, 0.1]])
    theta = torch.tensor([[0.5, -1.0, 1.5], [0.0, 0.0, 0.0], [1.9, -1.9, 0.3], [2.5, 0.0, 0.0]])
    lp = post.log_prob(theta, x=x)
    want = exact(theta, 0.6)
    assert torch.allclose(lp[:3], want[:3], atol=2e-5)
    assert lp[3] == float("-inf")
    # default x, 1-D theta, chunked evaluation, tolerances passed through
    post.set_default_x(x)
    assert torch.allclose(post.log_prob(theta[0]), want[:1], atol=2e-5)
    assert torch.allclose(post.log_prob(theta[:3], max_batch_size=2), want[:3], atol=2e-5)
    loose = post.log_prob(theta[:3], ode_kwargs=dict(atol=1e-2, rtol=1e-2))
    assert (loose - want[:3]).abs().max() > 1e-6 and torch.allclose(loose, want[:3], atol=1e-2)
    assert post.log_prob(torch.zeros(0, 3)).shape == (0,)


def test_iid_observations_and_refusals():
    est = LinearField()
    prior = Independent(Normal(torch.zeros(3), 2.0 * torch.ones(3)), 1)
    post = VectorFieldPosterior(est, prior, device="cpu")
    theta = torch.tensor([[0.5, -1.0, 1.5], [0.1, 0.2, -0.3]])
    xs = torch.tensor([[0.2, 0.1], [0.0, -0.1], [0.4, 0.4]])
    got = post.log_prob(theta, x=xs)
    want = sum(exact(theta, est.rate(xs[i])) for i in range(3)) - 2.0 * prior.log_prob(theta)
    assert torch.allclose(got, want, atol=5e-5)
    with pytest.raises(AssertionError):
        VectorFieldPosterior(est, None, device="cpu").log_prob(theta, x=xs)
    with pytest.raises(NotImplementedError):
        post.log_prob(theta, x=xs[:1], track_gradients=True)
    with pytest.raises(NotImplementedError):
        post.log_prob(theta, x=xs[:1], ode_kwargs=dict(exact=False))
    with pytest.raises(TypeError):
        post.log_prob(theta, x=xs[:1], ode_kwargs=dict(method="euler"))
    with pytest.raises(ValueError):
        post.log_prob(torch.zeros(2, 4), x=xs[:1])
    with pytest.raises(ValueError):
        post.log_prob(theta)          # no default x
