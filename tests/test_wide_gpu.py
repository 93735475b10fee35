"""hidden_features 65 ... 128 (nflows' ResidualNet takes any width, flow.py:333-349): the wide cooperative kernels
(csrc/nsf_coop_wide_kernel.h: two hidden m-tiles per wave, weights from L2) against the CPU oracle -- log_prob,
transform_to_noise, sampling for given noise, the inverse round trip, the training pass (per-row loss, flat parameter
gradient against fp64 autograd, d loss / d theta, row weights) and the fused step -- at every batch size, ragged rows
included."""
import pytest
import torch

from sbi_amd import _lib
from tests.helpers import hip_training_pass as _hip_pass, make_inputs, matched_pair, oracle_training_grad as _oracle_grad, \
    spline_knot_distances
from tests.parity_log import record

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(D=10, C=10, hidden_features=100),
    dict(D=10, C=10, hidden_features=128, num_transforms=3),
    dict(D=2, C=2, hidden_features=65, num_transforms=2),
    dict(D=5, C=20, hidden_features=96, num_transforms=2, num_blocks=1, num_bins=16),
    dict(D=16, C=32, hidden_features=80, num_transforms=2, num_bins=8),
    dict(D=7, C=3, hidden_features=128, num_transforms=2, num_blocks=3, num_bins=4),
    dict(D=5, C=50, hidden_features=100, num_transforms=2),                    # x-dim 33 ... 64: four context quads
    dict(D=10, C=64, hidden_features=128, num_transforms=2, num_blocks=1),
]


def _ids(c):
    return "-".join(f"{k}{v}" for k, v in c.items())


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_log_prob_and_noise_match_oracle(cfg):
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    lib, c = _lib.load(), est.net.hyper.c_config()
    assert lib.sbi_amd_nsf_image_kind(c, 7, 0) == 1 and lib.sbi_amd_nsf_image_kind(c, 65536, 0) == 1   # any batch size
    for what, (theta, x) in (("in-distribution", (theta_d[:777], x_d[:777])),
                             ("stress", make_inputs(1000, cfg["D"], cfg["C"]))):
        with torch.no_grad():
            ref = oracle.log_prob(theta, x)[0]
            ref64 = oracle.double().log_prob(theta.double(), x.double())[0]
            oracle.float()
        got = est.log_prob(theta.cuda(), x.cuda())[0].detach().cpu()
        assert torch.isfinite(got).all()
        e_o = (got - ref).abs().max().item()
        e_hip, e_ref = (got.double() - ref64).abs().max().item(), (ref.double() - ref64).abs().max().item()
        record("wide_log_prob", _ids(cfg) + " | " + what, max_abs_hip_vs_oracle32=e_o, max_abs_hip_vs_f64=e_hip,
               max_abs_oracle32_vs_f64=e_ref, max_abs_ref=ref.abs().max().item())
        assert e_o <= 1e-5 + 1e-5 * ref.abs().max().item(), (what, e_o)
        assert e_hip <= 2.0 * e_ref + 1e-5, (what, e_hip, e_ref)
    # ragged batches, one broadcast observation
    for n in (1, 15, 16, 17, 333):
        got = est.log_prob(theta_d[:n].cuda(), x_d[:1].cuda().expand(n, -1).contiguous())[0].cpu()
        with torch.no_grad():
            ref = oracle.log_prob(theta_d[:n], x_d[:1].expand(n, -1))[0]
        assert (got - ref).abs().max().item() <= 1e-5 + 1e-5 * ref.abs().max().item(), n


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_sampling_direction_matches_oracle_and_inverts_the_density_direction(cfg):
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    n = 500
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(n, cfg["D"], generator=g)
    x = x_d[:n]
    with torch.no_grad():
        ref, ref_ld = oracle.sample_from_noise(noise, x)
    got, got_ld = est.sample_from_noise(noise.cuda(), x.cuda(), with_logabsdet=True)
    e = (got.cpu() - ref).abs().max().item()
    e_ld = (got_ld.cpu() - ref_ld).abs().max().item()
    record("wide_sample", _ids(cfg), max_abs_hip_vs_oracle32=e, max_abs_logabsdet=e_ld, max_abs_ref=ref.abs().max().item())
    assert e <= 2e-5 * max(1.0, ref.abs().max().item()), e
    assert e_ld <= 2e-5 * max(1.0, ref_ld.abs().max().item()) + 1e-5, e_ld
    # round trip: theta -> noise (density direction) -> theta (sampling direction)
    th = theta_d[:n]
    z = est.inverse_transform(th.cuda(), x.cuda())
    back = est.sample_from_noise(z, x.cuda())
    assert (back.cpu() - th).abs().max().item() <= 2e-4 * max(1.0, th.abs().max().item())


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_training_pass_matches_fp64_autograd(cfg):
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    for n, weighted in ((333, True), (16, False), (1, False)):
        theta, x = theta_d[:n], x_d[:n]
        w = torch.full((n,), 1.0 / n)
        if weighted:
            g = torch.Generator().manual_seed(4)
            w = torch.rand(n, generator=g) / n
            w[::9] = 0.0                                    # zero-weight rows must contribute exactly nothing
        # the RQ spline is C1: a row whose spline input sits within two fp32 spacings of a knot (verified in fp64
        # through the oracle) has a two-valued gradient, and fp32 and fp64 may land on different sides -- such rows
        # get weight zero here (they then contribute exactly nothing in both implementations)
        near = spline_knot_distances(oracle, theta, x).min(1).values < 2.0
        assert int(near.sum()) <= 2
        w[near] = 0.0
        l64, g64, gth64, _ = _oracle_grad(oracle, est, theta, x, w)
        _, g32, _, _ = _oracle_grad(oracle, est, theta, x, w, double=False)
        l_c, g_c, gth_c, _ = _hip_pass(est, theta, x, w)
        assert torch.isfinite(g_c).all() and torch.isfinite(gth_c).all()
        scale = g64.abs().max().item()
        e_c = (g_c.double() - g64).abs().max().item() / scale
        e_o = (g32.double() - g64).abs().max().item() / scale
        e_l = (l_c.double() - l64).abs().max().item() / (1 + l64.abs().max().item())
        e_t = (gth_c.double() - gth64).abs().max().item() / max(gth64.abs().max().item(), 1e-12)
        record("wide_train", _ids(cfg) + f" | n={n}", grad_rel_hip_vs_f64=e_c, grad_rel_oracle32_vs_f64=e_o,
               loss_rel=e_l, grad_theta_rel=e_t)
        assert e_l <= 2e-5, (n, e_l)
        assert e_c <= max(2e-4, 6.0 * e_o), (n, e_c, e_o)
        assert e_t <= 2e-4, (n, e_t)
        # every parameter block individually (a wrong slab tile would hide behind the largest block's scale)
        for key, off, cnt, _ in est.net._slices():
            a, b = g_c[off : off + cnt].double(), g64[off : off + cnt]
            tol = 3e-4 * max(b.abs().max().item(), 1e-3 * scale) + 1e-9
            assert (a - b).abs().max().item() <= tol, (n, key)


def test_trains_end_to_end_at_hidden_100():
    """NPE with hidden_features = 100 (fused step on the wide kernels, device sampler): the loss goes down and the
    posterior of the 2-D linear-Gaussian task is recovered (C2ST against the analytic posterior)."""
    import warnings

    from torch.distributions import MultivariateNormal

    from sbi_amd.inference import NPE
    from sbi_amd.neural_nets import NSFConfig
    from sbi_amd.simulators.linear_gaussian import linear_gaussian, true_posterior_linear_gaussian_mvn_prior
    from sbi_amd.utils.metrics import c2st

    torch.manual_seed(0)
    dim, n = 2, 3000
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
    theta = prior.sample((n,)).cpu()
    x = linear_gaussian(theta, shift, cov)
    inf = NPE(prior=prior, density_estimator=NSFConfig(hidden_features=100, num_transforms=3), device="cuda",
              show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.append_simulations(theta, x).train(training_batch_size=500, max_num_epochs=60)
    s = inf.summary
    assert s["validation_loss"][-1] < s["validation_loss"][0] - 0.5
    post = inf.build_posterior(est)
    x_o = torch.zeros(1, dim)
    samples = post.sample((2000,), x=x_o.cuda(), show_progress_bars=False).cpu()
    target = true_posterior_linear_gaussian_mvn_prior(x_o[0], shift, cov, torch.zeros(dim), torch.eye(dim)).sample((2000,))
    score = c2st(samples, target).item()
    print("c2st at hidden 100:", score)
    assert 0.4 <= score <= 0.62


@pytest.mark.parametrize("cfg", [CONFIGS[0], CONFIGS[1], CONFIGS[4]], ids=_ids)
def test_training_pass_above_4096_rows_two_tiles_per_workgroup(cfg):
    """More than 4 096 rows: the backward pass contracts the weight gradients over two 16-row tiles per workgroup when
    its tiles fit LDS (half the partial slabs), one otherwise; ragged row count, the forward pass stays at one tile."""
    oracle, est, theta_d, x_d = matched_pair(n=5000, **cfg)
    n = 4999
    theta, x = theta_d[:n], x_d[:n]
    w = torch.full((n,), 1.0 / n)
    near = spline_knot_distances(oracle, theta, x).min(1).values < 2.0
    assert int(near.sum()) <= 8
    w[near] = 0.0
    l64, g64, gth64, _ = _oracle_grad(oracle, est, theta, x, w)
    l_c, g_c, gth_c, _ = _hip_pass(est, theta, x, w)
    scale = g64.abs().max().item()
    e_c = (g_c.double() - g64).abs().max().item() / scale
    e_l = (l_c.double() - l64).abs().max().item() / (1 + l64.abs().max().item())
    e_t = (gth_c.double() - gth64).abs().max().item() / max(gth64.abs().max().item(), 1e-12)
    record("wide_train", _ids(cfg) + f" | n={n}", grad_rel_hip_vs_f64=e_c, loss_rel=e_l, grad_theta_rel=e_t)
    assert e_l <= 2e-5 and e_c <= 3e-4 and e_t <= 3e-4, (e_l, e_c, e_t)
    for key, off, cnt, _ in est.net._slices():
        a, b = g_c[off : off + cnt].double(), g64[off : off + cnt]
        assert (a - b).abs().max().item() <= 3e-4 * max(b.abs().max().item(), 1e-3 * scale) + 1e-9, key


def test_lu_inverses_are_packed_for_the_sampling_direction_only():
    from sbi_amd.neural_nets.estimators.nsf_flow import packed_weights

    _, est, theta, x = matched_pair(D=4, C=3, hidden_features=100, num_transforms=2)
    net = est.net
    net.__dict__.pop("_packed_cache", None)
    net.__dict__.pop("_packed_images", None)
    packed_weights(net, rows=200, training=True)
    assert net.__dict__["_packed_images"] == 2          # the image the training pass reads, without U^-1 / L^-1
    est.log_prob(theta[:50].cuda(), x[:50].cuda())
    assert net.__dict__["_packed_images"] == 2
    a = est.sample_from_noise(torch.randn(50, 4, device="cuda"), x[:50].cuda())
    assert net.__dict__["_packed_images"] == 6          # + the inverses, packed once per weight version
    with torch.no_grad():
        net.flat_params.add_(0.01)
    b = est.sample_from_noise(torch.randn(50, 4, device="cuda"), x[:50].cuda())
    assert net.__dict__["_packed_images"] == 6 and torch.isfinite(a).all() and torch.isfinite(b).all()
    z = est.inverse_transform(b, x[:50].cuda())
    back = est.sample_from_noise(z, x[:50].cuda())
    assert (back - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())     # inverses of the NEW weights
