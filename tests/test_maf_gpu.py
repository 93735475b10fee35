"""GPU parity of the maf_rqs kernels (csrc/maf_kernel.h) through the C ABI against the CPU oracle
(oracle/maf_oracle.py) on identical weights and inputs: log_prob, sample for given noise, inverse_transform, the
flat parameter gradient and d loss / d theta of the fused training pass, plus end-to-end NPE with
density_estimator "maf_rqs" (tests/linearGaussian_snpe_test.py:155-200 lists it next to "nsf").
Tolerances as for the NSF path (tests/test_nsf_parity_gpu.py): 1e-5 norm-wise and "no further from fp64 than the
fp32 oracle is (x2)"."""
import warnings

import pytest
import torch
from torch.distributions import MultivariateNormal

from oracle.maf_oracle import MAFRQSOracle
from sbi_amd.neural_nets.net_builders.flow import build_maf_rqs
from tests.helpers import linear_gaussian_data, make_inputs
from tests.parity_log import record

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(D=10, C=10),
    dict(D=2, C=2),
    dict(D=4, C=7),
    dict(D=3, C=5, hidden_features=32, num_transforms=3, num_bins=8, num_blocks=1),
    dict(D=5, C=3, hidden_features=64, num_transforms=2, num_bins=5),
    dict(D=1, C=3, num_transforms=2),
    dict(D=16, C=12, num_transforms=2, num_bins=4, hidden_features=40, num_blocks=3),
    dict(D=7, C=4, num_bins=16, num_transforms=2, tail_bound=5.0),
]


def _ids(c):
    return "-".join(f"{k}{v}" for k, v in c.items())


def maf_pair(D, C, n=1000, perturb=0.05, seed=1, **kw):
    theta, x = linear_gaussian_data(n, D, C)
    torch.manual_seed(seed)
    oracle = MAFRQSOracle(theta, x, **kw)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for p in oracle.parameters():
            p.add_(perturb * torch.randn(p.shape, generator=g))
    est = build_maf_rqs(theta, x, **kw)
    est.net.load_nflows_state_dict(oracle.state_dict())
    return oracle, est.cuda(), theta, x


def oracle_flat_grad(oracle, est, dtype=torch.float32):
    named = dict(oracle.named_parameters())
    out = torch.zeros(est.net.flat_params.numel(), dtype=dtype)
    h = est.net.hyper
    kinds = {key: kind for key, _, kind in h.layer_entries()}
    for key, off, n, shape in est.net._slices():
        g = named["net." + key].grad
        sub = key.split(".", 3)[3]                      # autoregressive_net....
        m = h.mask(kinds[sub]) if kinds[sub] >= 0 else None
        out[off : off + n] = (g * m.to(g.dtype) if m is not None else g).reshape(-1)
    return out


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_log_prob_and_sample_match_oracle(cfg):
    oracle, est, theta_d, x_d = maf_pair(**cfg)
    D, C = cfg["D"], cfg["C"]
    for what, (theta, x) in (("in-distribution", (theta_d[:777], x_d[:777])), ("stress", make_inputs(2048, D, C))):
        with torch.no_grad():
            ref = oracle.log_prob(theta, x)[0]
            ref64 = oracle.double().log_prob(theta.double(), x.double())[0]
            oracle.float()
        got = est.log_prob(theta.cuda(), x.cuda())[0].cpu()
        assert torch.isfinite(got).all()
        e_hip, e_ref = (got.double() - ref64).abs().max().item(), (ref.double() - ref64).abs().max().item()
        record("maf_log_prob", _ids(cfg) + " | " + what, max_abs_hip_vs_oracle32=(got - ref).abs().max().item(),
               max_abs_hip_vs_f64=e_hip, max_abs_oracle32_vs_f64=e_ref, max_abs_ref=ref.abs().max().item())
        print(f"{what}: |hip-o32|={(got - ref).abs().max():.3e} |hip-f64|={e_hip:.3e} |o32-f64|={e_ref:.3e} "
              f"max|ref|={ref.abs().max():.1f}")
        if what == "in-distribution":
            assert (got - ref).abs().max() <= 1e-5 + 1e-5 * ref.abs().max()
        assert e_hip <= 2.0 * e_ref + 1e-5
    # sample = transform^-1(noise | x) for GIVEN noise; D conditioner passes per transform
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
    record("maf_sample", _ids(cfg), max_abs_hip_vs_oracle32=(got - ref).abs().max().item(), max_abs_hip_vs_f64=e_hip,
           max_abs_oracle32_vs_f64=e_ref, max_abs_ref=ref.abs().max().item())
    print(f"sample: |hip-o32|={(got - ref).abs().max():.3e} |hip-f64|={e_hip:.3e} |o32-f64|={e_ref:.3e}")
    assert e_hip <= 2.0 * e_ref + 1e-5
    assert (got_ld.double() - ref_ld64).abs().max() <= 2.0 * (ref_ld.double() - ref_ld64).abs().max() + 2e-5
    # round trip on the device
    back = est.inverse_transform(got.cuda(), x.cuda()).cpu()
    assert (back - noise).abs().max() <= 2e-4


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_training_pass_matches_autograd(cfg):
    from sbi_amd.neural_nets.estimators.maf_flow import maf_loss_fwd_bwd

    oracle, est, theta_d, x_d = maf_pair(**cfg)
    n = 333      # ragged: not a multiple of the 16-row wave tile
    theta, x = theta_d[:n], x_d[:n]
    w = torch.linspace(0.5, 1.5, n) / n
    oracle.double().zero_grad()
    th = theta.double().clone().requires_grad_(True)
    loss_ref = oracle.loss(th, x.double())
    (loss_ref * w.double()).sum().backward()
    gref = oracle_flat_grad(oracle, est, torch.float64)
    gth_ref = th.grad.clone()
    oracle.float()
    grad = torch.full_like(est.net.flat_params.data, float("nan"))
    from sbi_amd.neural_nets.estimators.maf_flow import MAFNet

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
    record("maf_train_grad", _ids(cfg), rel_grad_err_vs_f64=rel, worst_block_rel_err=worst, rel_grad_theta_err=e_th)
    print(f"grad rel {rel:.3e} worst block {worst:.3e} d/dtheta rel {e_th:.3e}")
    assert rel <= 2e-4 and e_th <= 3e-4
    # masked entries of the weight gradients are exactly zero
    h = est.net.hyper
    for (key, off, cnt, shape), (_, _, kind) in zip(est.net._slices(), h.layer_entries() * h.num_transforms):
        if kind in (0, 2, 3):
            assert (got[off : off + cnt].reshape(shape)[h.mask(kind) == 0] == 0).all(), key


def test_autograd_bridge_and_fused_step():
    from sbi_amd.inference.trainers.fused import FusedTrainStep

    oracle, est, theta_d, x_d = maf_pair(D=4, C=7)
    theta, x = theta_d[:200], x_d[:200]
    oracle.zero_grad()
    oracle.loss(theta, x).mean().backward()
    gref = oracle_flat_grad(oracle, est)
    est.zero_grad()
    est.loss(theta.cuda(), x.cuda()).mean().backward()
    assert (est.net.flat_params.grad.cpu() - gref).abs().max() <= 2e-4 * gref.abs().max()
    stepper = FusedTrainStep(est, lr=5e-4, clip_max_norm=5.0)
    stepper.loss_and_grad(theta.cuda(), x.cuda())
    assert (stepper.grad.cpu() - gref).abs().max() <= 2e-4 * gref.abs().max()
    first = stepper.step(theta.cuda(), x.cuda()).mean().item()
    for _ in range(40):
        last = stepper.step(theta.cuda(), x.cuda()).mean().item()
    assert last < first - 0.05


def test_full_size_log_prob_and_grad_65536():
    """BASELINE batch size against the oracle (chunked), theta-dim = x-dim = 10."""
    from sbi_amd.neural_nets.estimators.maf_flow import maf_loss_fwd_bwd

    oracle, est, _, _ = maf_pair(D=10, C=10)
    N, CH = 65536, 16384
    g = torch.Generator().manual_seed(0)
    theta = torch.randn(N, 10, generator=g) * (0.1**0.5)
    x = theta + (0.1**0.5) * torch.randn(N, 10, generator=g)
    oracle.double().zero_grad()
    ref = []
    for i in range(0, N, CH):
        l = oracle.loss(theta[i : i + CH].double(), x[i : i + CH].double())
        (l.sum() / N).backward()
        ref.append(l.detach())
    ref = torch.cat(ref)
    gref = oracle_flat_grad(oracle, est, torch.float64)
    oracle.float()
    grad = torch.empty_like(est.net.flat_params.data)
    losses, _ = maf_loss_fwd_bwd(est.net, theta.cuda(), x.cuda(), None, 1.0 / N, grad)
    e_l = (losses.cpu().double() - ref).abs().max().item()
    rel = (grad.cpu().double() - gref).abs().max().item() / gref.abs().max().item()
    record("maf_65536", "D10-C10", max_abs_loss_err_vs_f64=e_l, max_abs_loss=ref.abs().max().item(),
           rel_grad_err_vs_f64=rel)
    print(f"65536 rows: loss err {e_l:.3e} (max {ref.abs().max():.1f}) grad rel {rel:.3e}")
    assert e_l <= 1e-5 + 2e-5 * ref.abs().max().item()
    assert rel <= 1e-3          # knot-straddling rows (tests/test_parity_full_size_gpu.py) bound this from below


def test_npe_with_maf_rqs_recovers_the_linear_gaussian_posterior():
    from sbi_amd.inference import NPE
    from sbi_amd.neural_nets import MAFRQSConfig
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
    inf = NPE(prior=prior, density_estimator=MAFRQSConfig(), device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=100)
    assert inf._stepper is not None          # the fused device-resident step trained it
    post = inf.build_posterior().set_default_x(x_o)
    samples = post.sample((1000,), show_progress_bars=False).cpu()
    score = c2st(samples, target).item()
    print(f"maf_rqs NPE c2st={score:.3f} epochs={inf.summary['epochs_trained'][-1]}")
    record("c2st", "maf_rqs dim3 2.5k sims", c2st=score)
    assert 0.4 <= score <= 0.6
    lp = post.log_prob(samples[:5].cuda())
    assert torch.isfinite(lp).all()


def test_kernels_implement_the_other_reading_of_the_sqrt_hidden_question():
    """`scale_by_sqrt_hidden=True` (see tests/test_maf_oracle.py::test_both_readings_...): the HIP path with the
    flag set matches the oracle with the flag set -- log_prob, sample and the flat gradient -- so flipping the
    default, should a real nflows say so, needs no kernel work."""
    import dataclasses

    cfg = dict(D=5, C=4, hidden_features=32, num_transforms=3, num_bins=8)
    theta, x = linear_gaussian_data(1000, cfg["D"], cfg["C"])
    torch.manual_seed(1)
    oracle = MAFRQSOracle(theta, x, scale_by_sqrt_hidden=True, **{k: v for k, v in cfg.items() if k not in "DC"})
    g = torch.Generator().manual_seed(101)
    with torch.no_grad():
        for p in oracle.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    est = build_maf_rqs(theta, x, **{k: v for k, v in cfg.items() if k not in "DC"})
    est.net.hyper = dataclasses.replace(est.net.hyper, scale_by_sqrt_hidden=True)
    est.net.load_nflows_state_dict(oracle.state_dict())
    est = est.cuda()
    th, xx = theta[:600], x[:600]
    with torch.no_grad():
        ref = oracle.log_prob(th, xx)[0]
        plain = MAFRQSOracle(theta, x, **{k: v for k, v in cfg.items() if k not in "DC"})
        plain.load_state_dict(oracle.state_dict())
        other = plain.log_prob(th, xx)[0]
    got = est.log_prob(th.cuda(), xx.cuda())[0].cpu()
    assert (got - ref).abs().max() <= 1e-5 * (1 + ref.abs().max())
    assert (got - other).abs().max() > 1e-3                     # and it is NOT the unscaled reading
    noise = torch.randn(256, cfg["D"], generator=g)
    with torch.no_grad():
        s_ref = oracle.sample_from_noise(noise, xx[:256])[0]
    assert (est.sample_from_noise(noise.cuda(), xx[:256].cuda()).cpu() - s_ref).abs().max() < 1e-5
    from sbi_amd.neural_nets.estimators.maf_flow import maf_loss_fwd_bwd

    oracle.zero_grad()
    oracle.loss(th, xx).mean().backward()
    gref = oracle_flat_grad(oracle, est)
    grad = torch.empty_like(est.net.flat_params.data)
    maf_loss_fwd_bwd(est.net, th.cuda(), xx.cuda(), None, 1.0 / 600, grad)
    assert (grad.cpu() - gref).abs().max() <= 2e-4 * gref.abs().max()
