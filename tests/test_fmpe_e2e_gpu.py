"""FMPE end to end on the GPU: ODE sampling against a tight-tolerance solve of the oracle's vector field, and
the trainer + posterior on the linear-Gaussian task (analytic posterior known)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ode_sampling_matches_tight_solve_of_oracle_field():
    from scipy.integrate import solve_ivp

    from sbi_amd.samplers.ode_solvers import odeint_dopri5
    from tests.test_fmpe_gpu import make_pair

    oracle, est, theta, x, _, _ = make_pair(D=3, C=2, H=64, L=2)
    torch.manual_seed(5)
    eps = torch.randn(6, 3)
    x_o = x[:1]

    def rhs(t, y):
        with torch.no_grad():
            v = oracle.velocity(torch.tensor(y.reshape(6, 3), dtype=torch.float32), x_o,
                                torch.full((6,), float(t)))
        return v.double().numpy().reshape(-1)

    ref = solve_ivp(rhs, (1.0, 0.0), eps.double().numpy().reshape(-1), rtol=1e-8, atol=1e-9).y[:, -1].reshape(6, 3)
    got = odeint_dopri5(lambda t, y: est.ode_fn(y, x_o.cuda(), t), eps.cuda(), 1.0, 0.0).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())


def test_fmpe_trains_and_recovers_linear_gaussian_posterior():
    from torch.distributions import Independent, Normal

    from sbi_amd.inference import FMPE

    torch.manual_seed(0)
    D, n = 3, 4000
    sp, sn = 1.0, 0.5
    prior = Independent(Normal(torch.zeros(D, device="cuda"), sp * torch.ones(D, device="cuda")), 1)
    theta = prior.sample((n,))
    x = theta + sn * torch.randn_like(theta)
    inference = FMPE(prior=prior, device="cuda", show_progress_bars=False)
    est = inference.append_simulations(theta, x).train(training_batch_size=200, max_num_epochs=60)
    s = inference.summary
    assert s["validation_loss"][-1] < s["validation_loss"][0]
    posterior = inference.build_posterior(est)
    x_o = torch.tensor([[0.8, -0.4, 0.2]], device="cuda")
    samples = posterior.sample((4000,), x=x_o)
    assert samples.shape == (4000, D)
    k = sp**2 / (sp**2 + sn**2)
    mean_true = (x_o[0] * k).cpu()
    std_true = (sp**2 * sn**2 / (sp**2 + sn**2)) ** 0.5
    m, sd = samples.mean(0).cpu(), samples.std(0).cpu()
    print("posterior mean", m.tolist(), "true", mean_true.tolist(), "std", sd.tolist(), "true", std_true)
    assert (m - mean_true).abs().max() < 0.12
    assert ((sd - std_true).abs() / std_true).max() < 0.25
    # classifier two-sample test against exact posterior draws (the reference's criterion,
    # tests/linearGaussian_vector_field_test.py: c2st close to 0.5)
    from sbi_amd.utils.metrics import c2st

    exact = mean_true + std_true * torch.randn(4000, D)
    score = float(c2st(samples.cpu(), exact))
    print("c2st(FMPE, exact) =", score)
    assert score < 0.56     # measured 0.507; the north star asks for <= 0.55 on linear-Gaussian tasks
    # batched observations: (samples, batch, D)
    sb = posterior.sample_batched((50,), x=torch.stack([x_o[0], -x_o[0]]))
    assert sb.shape == (50, 2, D)
    assert (sb[:, 0].mean(0).cpu() - mean_true).abs().max() < 0.4
    # log-density of the trained flow (augmented probability-flow ODE) against the analytic Gaussian posterior
    lp = posterior.log_prob(samples[:500], x=x_o).cpu()
    true = Independent(Normal(mean_true, std_true * torch.ones(D)), 1).log_prob(samples[:500].cpu())
    print("log_prob: mean(flow - analytic)", (lp - true).mean().item(), "mean |.|", (lp - true).abs().mean().item())
    assert torch.isfinite(lp).all()
    assert (lp - true).mean().abs().item() < 0.25 and (lp - true).abs().mean().item() < 0.5
