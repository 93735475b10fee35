"""GPU parity of the zuko_nsf path (maf kernels, variant 1) through the C ABI against oracle/zuko_oracle.py on
identical weights and inputs: log_prob, sample for given noise, the fused training pass's gradients, NPE end to end
(tests/linearGaussian_snpe_test.py:155-200 lists "zuko_nsf")."""
import warnings

import pytest
import torch
from torch.distributions import MultivariateNormal

from oracle.zuko_oracle import ZukoNSFOracle
from sbi_amd.neural_nets.net_builders.flow import build_zuko_nsf
from tests.helpers import linear_gaussian_data, make_inputs
from tests.parity_log import record

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(D=10, C=10),
    dict(D=2, C=2),
    dict(D=4, C=7, num_transforms=3),
    dict(D=3, C=5, hidden_features=32, num_transforms=2, num_bins=8),
    dict(D=1, C=3, num_transforms=2),
    dict(D=16, C=20, num_transforms=2, num_bins=4, hidden_features=64),
    dict(D=5, C=3, hidden_features=[40, 40, 40], num_transforms=4, num_bins=16),
]


def _ids(c):
    return "-".join(f"{k}{v}" for k, v in c.items())


def zuko_pair(D, C, n=1000, perturb=0.05, seed=1, **kw):
    theta, x = linear_gaussian_data(n, D, C)
    torch.manual_seed(seed)
    oracle = ZukoNSFOracle(theta, x, **kw)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for p in oracle.parameters():
            p.add_(perturb * torch.randn(p.shape, generator=g))
    est = build_zuko_nsf(theta, x, **kw)
    est.net.load_zuko_state_dict(oracle.state_dict())
    return oracle, est.cuda(), theta, x


def oracle_flat_grad(oracle, est, dtype=torch.float64):
    named = dict(oracle.named_parameters())
    out = torch.zeros(est.net.flat_params.numel(), dtype=dtype)
    mask = est.net.mask_flat.cpu().to(dtype)
    for key, off, n, shape in est.net._slices():
        out[off : off + n] = named[key].grad.reshape(-1)
    return out * mask


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_log_prob_and_sample_match_oracle(cfg):
    oracle, est, theta_d, x_d = zuko_pair(**cfg)
    D, C = cfg["D"], cfg["C"]
    for what, (theta, x) in (("in-distribution", (theta_d[:777], x_d[:777])), ("stress", make_inputs(2048, D, C))):
        with torch.no_grad():
            ref = oracle.log_prob(theta, x)[0]
            ref64 = oracle.double().log_prob(theta.double(), x.double())[0]
            oracle.float()
        got = est.log_prob(theta.cuda(), x.cuda())[0].cpu()
        assert torch.isfinite(got).all()
        e_hip, e_ref = (got.double() - ref64).abs().max().item(), (ref.double() - ref64).abs().max().item()
        record("zuko_log_prob", _ids(cfg) + " | " + what, max_abs_hip_vs_oracle32=(got - ref).abs().max().item(),
               max_abs_hip_vs_f64=e_hip, max_abs_oracle32_vs_f64=e_ref, max_abs_ref=ref.abs().max().item())
        print(f"{what}: |hip-o32|={(got - ref).abs().max():.3e} |hip-f64|={e_hip:.3e} |o32-f64|={e_ref:.3e} "
              f"max|ref|={ref.abs().max():.1f}")
        if what == "in-distribution":
            assert (got - ref).abs().max() <= 1e-5 + 1e-5 * ref.abs().max()
        assert e_hip <= 2.0 * e_ref + 1e-5
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(1000, D, generator=g)
    x = x_d[:1000]
    with torch.no_grad():
        ref, ref_ld = oracle.sample_from_noise(noise, x)
        ref64, ref_ld64 = oracle.double().sample_from_noise(noise.double(), x.double())
        oracle.float()
    got, got_ld = est.sample_from_noise(noise.cuda(), x.cuda(), with_logabsdet=True)
    got, got_ld = got.cpu(), got_ld.cpu()
    e_hip, e_ref = (got.double() - ref64).abs().max().item(), (ref.double() - ref64).abs().max().item()
    record("zuko_sample", _ids(cfg), max_abs_hip_vs_oracle32=(got - ref).abs().max().item(), max_abs_hip_vs_f64=e_hip,
           max_abs_oracle32_vs_f64=e_ref, max_abs_ref=ref.abs().max().item())
    print(f"sample: |hip-o32|={(got - ref).abs().max():.3e} |hip-f64|={e_hip:.3e} |o32-f64|={e_ref:.3e}")
    assert e_hip <= 2.0 * e_ref + 1e-5
    assert (got_ld.double() - ref_ld64).abs().max() <= 2.0 * (ref_ld.double() - ref_ld64).abs().max() + 2e-5
    back = est.inverse_transform(got.cuda(), x.cuda()).cpu()
    assert (back - noise).abs().max() <= 2e-4


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_training_pass_matches_autograd(cfg):
    from sbi_amd.neural_nets.estimators.maf_flow import maf_loss_fwd_bwd

    oracle, est, theta_d, x_d = zuko_pair(**cfg)
    n = 333
    theta, x = theta_d[:n], x_d[:n]
    w = torch.linspace(0.5, 1.5, n) / n
    oracle.double().zero_grad()
    th = theta.double().clone().requires_grad_(True)
    loss_ref = oracle.loss(th, x.double())
    (loss_ref * w.double()).sum().backward()
    gref = oracle_flat_grad(oracle, est)
    gth_ref = th.grad.clone()
    oracle.float()
    grad = torch.full_like(est.net.flat_params.data, float("nan"))
    ws = torch.full((est.net.train_workspace_floats(n),), float("nan"), device="cuda")
    losses, gth = maf_loss_fwd_bwd(est.net, theta.cuda(), x.cuda(), w.cuda(), 0.0, grad, want_grad_theta=True,
                                   workspace=ws)
    torch.cuda.synchronize()
    got = grad.cpu().double()
    assert torch.isfinite(got).all() and torch.isfinite(gth).all()
    assert (losses.cpu().double() - loss_ref.detach()).abs().max() <= 1e-5 + 1e-5 * loss_ref.abs().max()
    scale = gref.abs().max().item()
    rel = (got - gref).abs().max().item() / scale
    worst = 0.0
    for key, off, cnt, _ in est.net._slices():
        a, b = got[off : off + cnt], gref[off : off + cnt]
        e = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-3 * scale)
        worst = max(worst, e)
        assert e <= 3e-4, f"{key}: {e:.3e}"
    e_th = (gth.cpu().double() - gth_ref).abs().max().item() / gth_ref.abs().max().item()
    record("zuko_train_grad", _ids(cfg), rel_grad_err_vs_f64=rel, worst_block_rel_err=worst, rel_grad_theta_err=e_th)
    print(f"grad rel {rel:.3e} worst block {worst:.3e} d/dtheta rel {e_th:.3e}")
    assert rel <= 2e-4 and e_th <= 3e-4
    assert (got[est.net.mask_flat.cpu() == 0] == 0).all()       # masked weights get exactly zero gradient


def test_npe_with_zuko_nsf_recovers_the_linear_gaussian_posterior():
    from sbi_amd.inference import NPE
    from sbi_amd.neural_nets import ZukoNSFConfig
    from sbi_amd.simulators.linear_gaussian import linear_gaussian, true_posterior_linear_gaussian_mvn_prior
    from sbi_amd.utils.metrics import c2st

    dim, n = 3, 3000
    torch.manual_seed(0)
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
    theta = prior.sample((n,)).cpu()
    x = linear_gaussian(theta, shift, cov)
    x_o = torch.zeros(1, dim)
    target = true_posterior_linear_gaussian_mvn_prior(x_o, shift, cov, torch.zeros(dim), torch.eye(dim)).sample((1000,))
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator=ZukoNSFConfig(), device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=100)
    assert inf._stepper is not None
    post = inf.build_posterior().set_default_x(x_o)
    samples = post.sample((1000,), show_progress_bars=False).cpu()
    score = c2st(samples, target).item()
    print(f"zuko_nsf NPE c2st={score:.3f} epochs={inf.summary['epochs_trained'][-1]}")
    record("c2st", "zuko_nsf dim3 3k sims", c2st=score)
    assert 0.4 <= score <= 0.6
