"""End-to-end NPE on the GPU path (BASELINE configs 1 and 2 plumbing): train with the
fused HIP step, build a DirectPosterior, sample, and compare with the analytic
linear-Gaussian posterior by C2ST (reference thresholds: tests/linearGaussian_snpe_test.py:53-200,
sbi/utils/metrics.py:167-175)."""

import warnings

import pytest
import torch
from torch.distributions import MultivariateNormal

from sbi_amd.inference import NPE
from sbi_amd.neural_nets import NSFConfig
from sbi_amd.simulators.linear_gaussian import (diagonal_linear_gaussian, linear_gaussian,
                                                true_posterior_linear_gaussian_mvn_prior)
from sbi_amd.utils.metrics import c2st
from sbi_amd.utils.torchutils import BoxUniform

pytestmark = pytest.mark.gpu


def test_cfg1_npe_nsf_linear_gaussian_dim2_c2st():
    """theta-dim 2, likelihood shift -1, cov 0.3 I, prior N(0, I), x_o = 0 (linearGaussian_snpe_test.py:60-92)."""
    dim, n = 2, 2500
    torch.manual_seed(0)
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
    theta = prior.sample((n,)).cpu()
    x = linear_gaussian(theta, shift, cov)
    x_o = torch.zeros(1, dim)
    target = true_posterior_linear_gaussian_mvn_prior(x_o, shift, cov, torch.zeros(dim), torch.eye(dim)).sample((1000,))
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator=NSFConfig(), device="cuda", show_progress_bars=False)
    est = inf.append_simulations(theta, x).train(training_batch_size=100)
    assert all(p.grad is None for p in est.parameters())
    post = inf.build_posterior().set_default_x(x_o)
    samples = post.sample((1000,), show_progress_bars=False).cpu()
    score = c2st(samples, target).item()
    print(f"cfg1 c2st={score:.3f} epochs={inf.summary['epochs_trained'][-1]}")
    assert 0.4 <= score <= 0.6
    # density agrees with the analytic posterior on its own samples (D-KL proxy)
    true_post = true_posterior_linear_gaussian_mvn_prior(x_o, shift, cov, torch.zeros(dim), torch.eye(dim))
    dkl = (true_post.log_prob(target) - post.log_prob(target.cuda()).cpu()).mean().item()
    print("monte-carlo D_KL(true || npe) =", dkl)
    assert dkl < 0.15


def test_uniform_prior_rejection_and_leakage_normalisation():
    dim, n = 2, 2500
    torch.manual_seed(0)
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
    prior = BoxUniform(-2.0 * torch.ones(dim), 2.0 * torch.ones(dim), device="cuda")
    theta = prior.sample((n,)).cpu()
    x = linear_gaussian(theta, shift, cov)
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator=NSFConfig(), device="cuda", show_progress_bars=False)
    inf.append_simulations(theta, x).train(training_batch_size=100)
    post = inf.build_posterior().set_default_x(torch.zeros(1, dim))
    s = post.sample((2000,), show_progress_bars=False)
    assert bool(prior.support.check(s).all())
    outside = torch.tensor([[2.5, 0.0]], device="cuda")
    assert post.log_prob(outside).item() == float("-inf")
    acc = post.leakage_correction(torch.zeros(1, dim, device="cuda"))
    assert 0.0 < acc.item() <= 1.0
    lp_n = post.log_prob(s[:10])
    lp_u = post.log_prob(s[:10], norm_posterior=False)
    assert torch.allclose(lp_n, lp_u - torch.log(acc), atol=1e-5)


def test_cfg2_dim10_posterior_c2st():
    """10-D task of tests/mini_sbibm/gaussian_linear.py: prior N(0, 0.1 I), x = theta + sqrt(0.1) eps;
    posterior N(x_o/2, 0.05 I).  Gate of the north_star: C2ST <= 0.55 -- asserted here for the accuracy
    configuration (100 000 simulations, batch 1 000, sbi's early stopping); tools/c2st_report.py reports the same
    number over three observations with 10 000 samples, and for the batch-65 536 benchmark configuration."""
    dim, n = 10, 100000
    torch.manual_seed(0)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), 0.1 * torch.eye(dim, device="cuda"))
    theta = prior.sample((n,)).cpu()
    x = diagonal_linear_gaussian(theta, std=0.1**0.5)
    torch.manual_seed(1)
    prior_cpu = MultivariateNormal(torch.zeros(dim), 0.1 * torch.eye(dim))
    x_o = diagonal_linear_gaussian(prior_cpu.sample((1,)), std=0.1**0.5)
    inf = NPE(prior=prior, density_estimator=NSFConfig(), device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=1000)
    post = inf.build_posterior().set_default_x(x_o)
    samples = post.sample((5000,), show_progress_bars=False).cpu()
    target = true_posterior_linear_gaussian_mvn_prior(x_o, torch.zeros(dim), 0.1 * torch.eye(dim), torch.zeros(dim),
                                                      0.1 * torch.eye(dim)).sample((5000,))
    score = c2st(samples, target).item()
    print(f"cfg2 c2st={score:.3f} epochs={inf.summary['epochs_trained'][-1]} "
          f"val={inf.summary['best_validation_loss'][-1]:.3f}")
    from tests.parity_log import record

    record("c2st", "cfg2 accuracy config (100k sims, batch 1000, 5000 samples)", c2st=score,
           epochs=inf.summary["epochs_trained"][-1])
    assert 0.45 <= score <= 0.55, "north_star gate: posterior C2ST <= 0.55 on linear-Gaussian"


def test_sample_1m_draws_direct_posterior():
    """BASELINE config 4 plumbing: 1M draws in one accept/reject pass on the device."""
    from sbi_amd.inference import DirectPosterior
    from sbi_amd.neural_nets.net_builders.flow import build_nsf
    from tests.helpers import linear_gaussian_data

    theta, x = linear_gaussian_data(2000, 10, 10)
    torch.manual_seed(1)
    est = build_nsf(theta, x).cuda()
    prior = BoxUniform(-3.0 * torch.ones(10), 3.0 * torch.ones(10), device="cuda")
    post = DirectPosterior(est, prior, device="cuda").set_default_x(torch.zeros(1, 10))
    s = post.sample((1_000_000,), max_sampling_batch_size=1_000_000, show_progress_bars=False)
    assert s.shape == (1_000_000, 10) and bool(prior.support.check(s).all())


@pytest.mark.parametrize("sample_with", ["direct", "rejection", "mcmc"])
def test_device_posterior_survives_pickling_with_identical_seeded_samples(sample_with):
    """tests/save_and_load_test.py:23-45 on a cuda posterior: `manual_seed(0); sample((3,))` before and after a pickle
    round trip give the same three draws (MCMC redraws a fresh chain by design, #1291: shape only), and pickling does
    not replace attributes of the posterior it serialised."""
    import pickle
    import warnings

    from sbi_amd.inference import NPE
    from tests.helpers import linear_gaussian_data

    D = 3
    theta, x = linear_gaussian_data(600, D, D)
    prior = BoxUniform(-3.0 * torch.ones(D), 3.0 * torch.ones(D), device="cuda")
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator="nsf", device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=100, max_num_epochs=3)
        kw = dict(mcmc_parameters=dict(num_chains=4, warmup_steps=5, thin=1)) if sample_with == "mcmc" else {}
        post = inf.build_posterior(sample_with=sample_with, **kw).set_default_x(torch.zeros(1, D))
    torch.manual_seed(0)
    expected = post.sample((3,), show_progress_bars=False)
    attributes = {name: id(value) for name, value in vars(post).items()}
    reloaded = pickle.loads(pickle.dumps(post))
    assert {name: id(value) for name, value in vars(post).items()} == attributes, \
        "pickling replaced attributes on the posterior it serialized"
    torch.manual_seed(0)
    samples = reloaded.sample((3,), show_progress_bars=False)
    assert samples.shape == (3, D) and samples.is_cuda
    if sample_with != "mcmc":
        assert torch.equal(samples, expected), (samples, expected)
    if sample_with == "direct":
        assert torch.equal(post.log_prob(expected), reloaded.log_prob(expected))


def test_reference_ci_scenario_nsf_npe_c_dim4_wall_time():
    """tests/linearGaussian_snpe_test.py:156-200 with density_estimator="nsf", NPE_C: theta-dim 4, 2 500
    simulations, batch 100, train to convergence, 1 000 posterior samples, C2ST near chance.  The reference's CI
    records 41.7 s for this test on a GitHub CPU runner (.test_durations:2075)."""
    import time

    dim, n = 4, 2500
    torch.manual_seed(0)
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
    x_o = torch.zeros(1, dim)
    target = true_posterior_linear_gaussian_mvn_prior(x_o, shift, cov, torch.zeros(dim), torch.eye(dim)).sample((1000,))
    theta = prior.sample((n,)).cpu()
    x = linear_gaussian(theta, shift, cov)
    torch.manual_seed(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inf = NPE(prior=prior, density_estimator=NSFConfig(), device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.append_simulations(theta, x).train(training_batch_size=100)
    from sbi_amd.inference import DirectPosterior

    post = DirectPosterior(prior=prior, posterior_estimator=est).set_default_x(x_o)
    samples = post.sample((1000,), show_progress_bars=False).cpu()
    wall = time.perf_counter() - t0
    score = c2st(samples, target).item()
    print(f"reference CI scenario: wall {wall:.2f} s (reference CI: 41.7 s), epochs {inf.summary['epochs_trained'][-1]}, "
          f"c2st {score:.3f}")
    assert 0.4 <= score <= 0.6


@pytest.mark.parametrize("stop_kind", ["early_stopping", "max_epochs"])
def test_pipelined_epoch_loop_takes_the_eager_loops_decisions(stop_kind, monkeypatch):
    """NPE.train() enqueues epoch e+1 before it reads epoch e's losses (one pinned read per epoch, one epoch
    late) and throws the speculative epoch away when the reference's rule says stop.  Same seeds => the same
    loss history, epoch count, best validation loss, returned weights and optimizer state as the loop that syncs
    every epoch (SBI_AMD_EAGER_EPOCH_SYNC=1)."""
    dim, n = 2, 700
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)

    def run(eager: bool):
        monkeypatch.setenv("SBI_AMD_EAGER_EPOCH_SYNC", "1" if eager else "0")
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
        theta = prior.sample((n,)).cpu()
        x = linear_gaussian(theta, shift, cov)
        torch.manual_seed(1)
        inf = NPE(prior=prior, density_estimator=NSFConfig(num_transforms=2), device="cuda", show_progress_bars=False)
        kw = dict(stop_after_epochs=3) if stop_kind == "early_stopping" else dict(max_num_epochs=7)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est = inf.append_simulations(theta, x).train(training_batch_size=100, learning_rate=5e-3, **kw)
        return inf, est

    a, est_a = run(eager=True)
    b, est_b = run(eager=False)
    assert a.summary["epochs_trained"] == b.summary["epochs_trained"]
    assert a.summary["validation_loss"] == b.summary["validation_loss"]
    assert a.summary["training_loss"] == b.summary["training_loss"]
    assert a.summary["best_validation_loss"] == b.summary["best_validation_loss"]
    assert torch.equal(est_a.net.flat_params, est_b.net.flat_params)
    assert torch.equal(a._stepper.exp_avg, b._stepper.exp_avg) and a._stepper.step_count == b._stepper.step_count
    if stop_kind == "early_stopping":
        assert a.summary["epochs_trained"][-1] < 200


def test_pipelined_loop_puts_back_the_last_finite_epoch_before_raising_on_nan(monkeypatch):
    """The losses of epoch e are read after epoch e + 1 was enqueued: when they turn out non-finite, the weights and
    the optimizer state must be the ones after the last epoch whose losses were finite, not whatever the non-finite
    steps left behind."""
    from sbi_amd.inference.trainers.fused import FusedTrainStep

    # (the injection below is host code inside `step`: it needs the eager loop -- a captured epoch replays kernels, not
    #  Python; the read-back / restore logic of `finish_epoch` under test is shared by both)
    dim, n = 2, 400
    torch.manual_seed(0)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
    theta = prior.sample((n,)).cpu()
    x = linear_gaussian(theta, -1.0 * torch.ones(dim), 0.3 * torch.eye(dim))
    real_step = FusedTrainStep.step
    seen = {"calls": 0, "good": None}

    def step(self, th, xx, **kw):      # one training batch per epoch: call k is epoch k - 1
        seen["calls"] += 1
        losses = real_step(self, th, xx, **kw)
        if seen["calls"] <= 3:
            seen["good"] = (self.net.flat_params.data.clone(), self.exp_avg.clone(), self.step_count)
            return losses
        self.net.flat_params.data.fill_(float("nan"))
        return losses * float("nan")

    monkeypatch.setattr(FusedTrainStep, "step", step)
    inf = NPE(prior=prior, density_estimator=NSFConfig(num_transforms=2), device="cuda", show_progress_bars=False)
    with pytest.raises(AssertionError, match="NaN/Inf"):
        inf.append_simulations(theta, x).train(training_batch_size=360, max_num_epochs=50)
    assert seen["calls"] >= 4
    params, exp_avg, steps = seen["good"]
    assert torch.isfinite(inf._stepper.net.flat_params).all()
    assert torch.equal(inf._stepper.net.flat_params.data, params)
    assert torch.equal(inf._stepper.exp_avg, exp_avg) and inf._stepper.step_count == steps
    assert len(inf.summary["training_loss"]) == 3        # epochs 0..2 were recorded, the non-finite one was not
    assert all(d > 0 for d in inf.summary["epoch_durations_sec"])
