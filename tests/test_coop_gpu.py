"""The cooperative small-batch kernels (csrc/nsf_coop.h: four waves per 16-row tile, register-fed MFMAs from the
fragment-ordered image, one backward launch for all transforms) through the C ABI:

* against the CPU oracle (log_prob, per-row loss, flat parameter gradient vs fp64 autograd, d loss / d theta,
  d loss / d embedded x, row weights, one broadcast x_o);
* against the throughput kernels on the same inputs (two independent HIP implementations of one function);
* ragged row counts 1 ... 8 192 across both workgroup shapes (one and two 16-row tiles per workgroup), with the
  workspace poisoned with NaN (nothing the kernels do not write themselves may reach a result);
* which calls take them (`sbi_amd_nsf_image_kind`), the image bookkeeping of `packed_weights`, run-to-run determinism.
"""
import pytest
import torch

from sbi_amd import _lib
from sbi_amd.neural_nets.estimators.nsf_flow import packed_weights
from tests.helpers import hip_training_pass as _hip_pass, make_inputs, matched_pair, oracle_training_grad as _oracle_grad
from tests.parity_log import record

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(D=10, C=10),
    dict(D=2, C=2),
    dict(D=3, C=5, hidden_features=32, num_transforms=3, num_blocks=1),
    dict(D=5, C=4, num_bins=16, num_transforms=2),
    dict(D=6, C=2, num_bins=5, hidden_features=40, num_transforms=3),
    dict(D=4, C=3, num_bins=4, num_transforms=2),
    dict(D=7, C=9, num_bins=8, num_transforms=2, num_blocks=3),
    dict(D=16, C=32, num_transforms=2),
    dict(D=15, C=20, num_transforms=3),
    dict(D=10, C=10, hidden_features=64, num_transforms=2),
    dict(D=9, C=17, hidden_features=64, num_transforms=2, num_blocks=4, num_bins=8),
]


def _ids(c):
    return "-".join(f"{k}{v}" for k, v in c.items())


class family:
    """with family("throughput"): ... -- route small calls to one kernel family for the duration."""

    def __init__(self, which):
        self.rows = 12288 if which == "cooperative" else 0

    def __enter__(self):
        self.prev = _lib.load().sbi_amd_nsf_set_coop_max_rows(self.rows)

    def __exit__(self, *a):
        _lib.load().sbi_amd_nsf_set_coop_max_rows(self.prev)


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_image_kind_and_log_prob_match_oracle_and_throughput_kernels(cfg):
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    lib, c = _lib.load(), est.net.hyper.c_config()
    assert lib.sbi_amd_nsf_image_kind(c, 200, 0) == 1 and lib.sbi_amd_nsf_image_kind(c, 200, 1) == 1
    assert lib.sbi_amd_nsf_image_kind(c, 65536, 0) == 0 and lib.sbi_amd_nsf_image_kind(c, 0, 0) == 0
    for what, (theta, x) in (("in-distribution", (theta_d[:777], x_d[:777])),
                             ("stress", make_inputs(2048, cfg["D"], cfg["C"]))):
        with torch.no_grad():
            ref = oracle.log_prob(theta, x)[0]
            ref64 = oracle.double().log_prob(theta.double(), x.double())[0]
            oracle.float()
        with family("cooperative"):
            got = est.log_prob(theta.cuda(), x.cuda())[0]
            noise = est.inverse_transform(theta.cuda(), x.cuda())
        with family("throughput"), torch.no_grad():
            try:
                thr = est.log_prob(theta.cuda(), x.cuda())[0].cpu()
                noise_thr = est.inverse_transform(theta.cuda(), x.cuda()).cpu()
            except RuntimeError:          # (a weight image that does not fit the throughput kernels' LDS budget)
                thr, noise_thr = None, None
        got = got.detach().cpu()
        assert torch.isfinite(got).all()
        e_o = (got - ref).abs().max().item()
        e_t = -1.0 if thr is None else (got - thr).abs().max().item()
        e_hip, e_ref = (got.double() - ref64).abs().max().item(), (ref.double() - ref64).abs().max().item()
        record("coop_log_prob", _ids(cfg) + " | " + what, max_abs_coop_vs_oracle32=e_o, max_abs_coop_vs_throughput=e_t,
               max_abs_coop_vs_f64=e_hip, max_abs_oracle32_vs_f64=e_ref, max_abs_ref=ref.abs().max().item())
        assert e_o <= 1e-5 + 1e-5 * ref.abs().max().item(), (what, e_o)
        assert e_hip <= 2.0 * e_ref + 1e-5, (what, e_hip, e_ref)
        if thr is not None:
            assert e_t <= 1e-5 + 2e-5 * ref.abs().max().item(), (what, e_t)
            assert (noise.cpu() - noise_thr).abs().max() <= 1e-4, "transform_to_noise differs between the families"


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_training_pass_matches_fp64_autograd_and_throughput_kernels(cfg):
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    n = 333
    theta, x = theta_d[:n], x_d[:n]
    g = torch.Generator().manual_seed(4)
    w = torch.rand(n, generator=g) / n
    w[::9] = 0.0                                        # zero-weight rows must contribute exactly nothing
    l64, g64, gth64, gx64 = _oracle_grad(oracle, est, theta, x, w)
    _, g32, _, _ = _oracle_grad(oracle, est, theta, x, w, double=False)
    with family("cooperative"):
        l_c, g_c, gth_c, gx_c = _hip_pass(est, theta, x, w, want_gx=True)
    with family("throughput"):
        try:
            l_t, g_t, gth_t, _ = _hip_pass(est, theta, x, w)
        except RuntimeError:                            # shapes only the cooperative kernels train at this size
            l_t = g_t = gth_t = None
    assert torch.isfinite(g_c).all() and torch.isfinite(gth_c).all() and torch.isfinite(gx_c).all()
    scale = g64.abs().max().item()
    e_c = (g_c.double() - g64).abs().max().item() / scale
    e_o = (g32.double() - g64).abs().max().item() / scale
    e_l = (l_c.double() - l64).abs().max().item() / (1 + l64.abs().max().item())
    e_th = (gth_c.double() - gth64).abs().max().item() / gth64.abs().max().item()
    e_x = (gx_c.double() - gx64).abs().max().item() / gx64.abs().max().item()
    e_t = None if g_t is None else (g_c - g_t).abs().max().item() / scale
    record("coop_train", _ids(cfg), rel_grad_err_coop_vs_f64=e_c, rel_grad_err_oracle32_vs_f64=e_o, rel_loss_err=e_l,
           rel_grad_theta_err=e_th, rel_grad_x_err=e_x, rel_grad_coop_vs_throughput=e_t if e_t is not None else -1.0)
    print(f"coop grad vs f64 {e_c:.2e} (oracle32 {e_o:.2e}), vs throughput {e_t}, dtheta {e_th:.2e}, dx {e_x:.2e}")
    assert e_l <= 1e-5
    assert e_c <= 2e-5 + 4 * e_o, f"flat gradient off by {e_c} of max|grad| (fp32 oracle: {e_o})"
    assert e_th <= 5e-5 and e_x <= 5e-5
    for key, off, cnt, _ in est.net._slices():          # per block, so a small block cannot hide behind a large one
        a, b = g_c[off : off + cnt].double(), g64[off : off + cnt]
        assert (a - b).abs().max().item() <= 2e-4 * max(b.abs().max().item(), 1e-3 * scale) + 1e-7, key
    if g_t is not None:
        assert e_t <= 1e-4 and (l_c - l_t).abs().max() <= 1e-4


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 31, 32, 33, 200, 1000, 4096, 4097, 4113, 8192])
def test_ragged_row_counts_both_workgroup_shapes(n):
    """n <= 4 096: one 16-row tile per workgroup; beyond: two (the second tile of the last workgroup may lie
    entirely past the last row: 4 097 / 4 113)."""
    cfg = dict(D=5, C=3, hidden_features=32, num_transforms=3, num_bins=8)
    oracle, est, theta_d, x_d = matched_pair(n=max(n, 1000), **cfg)
    theta, x = theta_d[:n], x_d[:n]
    l64, g64, gth64, _ = _oracle_grad(oracle, est, theta, x)
    with family("cooperative"):
        l_c, g_c, gth_c, _ = _hip_pass(est, theta, x)
        lp = est.log_prob(theta.cuda(), x.cuda())[0].cpu()
        l_c2, g_c2, _, _ = _hip_pass(est, theta, x)
    assert torch.equal(g_c, g_c2) and torch.equal(l_c, l_c2), "the cooperative pass is not run-to-run deterministic"
    scale = g64.abs().max().item()
    assert (l_c.double() - l64).abs().max() <= 1e-5 * (1 + l64.abs().max())
    assert (lp.double() + l64).abs().max() <= 1e-5 * (1 + l64.abs().max())
    assert (g_c.double() - g64).abs().max().item() <= 1e-4 * scale
    assert (gth_c.double() - gth64).abs().max() <= 1e-4 * gth64.abs().max()


def test_one_broadcast_condition_and_atoms_major_rows():
    """x_rows < n (the samplers' single x_o; the atomic loss' atoms-major layout: row r is conditioned on x[r % B])."""
    cfg = dict(D=4, C=6, num_transforms=3)
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    n, B = 600, 200
    theta = theta_d[:n]
    for xr in (1, B):
        x = x_d[:xr]
        xe = x.repeat(n // xr, 1)
        l64, g64, gth64, _ = _oracle_grad(oracle, est, theta, xe)
        with family("cooperative"):
            l_c, g_c, gth_c, _ = _hip_pass(est, theta, x)
        assert (l_c.double() - l64).abs().max() <= 1e-5 * (1 + l64.abs().max())
        assert (g_c.double() - g64).abs().max() <= 1e-4 * g64.abs().max()
        assert (gth_c.double() - gth64).abs().max() <= 1e-4 * gth64.abs().max()


def test_packed_weights_repacks_only_the_image_a_call_reads():
    _, est, theta, x = matched_pair(D=4, C=3, num_transforms=2)
    net = est.net
    net.__dict__.pop("_packed_cache", None)
    net.__dict__.pop("_packed_images", None)
    with family("cooperative"):
        packed_weights(net, rows=200, training=True)
        assert net.__dict__["_packed_images"] == 2
        packed_weights(net, rows=65536)
        assert net.__dict__["_packed_images"] == 3
        with torch.no_grad():
            net.flat_params.add_(0.01)
        a = est.log_prob(theta[:100].cuda(), x[:100].cuda())[0]
        assert net.__dict__["_packed_images"] == 2
        b = est.sample_from_noise(torch.randn(100, 4, device="cuda"), x[:100].cuda())
        # a small sampling call reads the cooperative image + its LU inverses (bit 4); a large one the throughput
        # image + its inverses (bit 8)
        assert net.__dict__["_packed_images"] == 6 and torch.isfinite(a).all() and torch.isfinite(b).all()
        est.sample_from_noise(torch.randn(20000, 4, device="cuda"), x[:1].cuda())
        assert net.__dict__["_packed_images"] == 15
    with family("throughput"):
        c = est.log_prob(theta[:100].cuda(), x[:100].cuda())[0]
    assert (a - c).abs().max() <= 1e-4


def test_npe_default_batch_trains_on_the_cooperative_kernels():
    """sbi's default training_batch_size = 200 (npe_base.py:301-316) end to end: NPE.train -> DirectPosterior -> C2ST
    against the analytic posterior of the linear-Gaussian task (tests/linearGaussian_snpe_test.py plumbing)."""
    import warnings

    from torch.distributions import MultivariateNormal

    from sbi_amd.inference import NPE
    from sbi_amd.utils.metrics import c2st

    torch.manual_seed(0)
    D = 3
    prior = MultivariateNormal(torch.zeros(D, device="cuda"), torch.eye(D, device="cuda"))
    theta = prior.sample((4000,)).cpu()
    x = theta + 0.5 * torch.randn_like(theta)
    inf = NPE(prior=prior, density_estimator="nsf", device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x)
        inf.train(training_batch_size=200, max_num_epochs=60)
        post = inf.build_posterior()
    x_o = torch.full((1, D), 0.4)
    s = post.sample((2000,), x=x_o.cuda(), show_progress_bars=False).cpu()
    cov = torch.eye(D) * (0.25 / 1.25)
    ref = MultivariateNormal(x_o[0] / 1.25, cov).sample((2000,))
    score = float(c2st(s, ref))
    record("coop_npe_c2st", "D3-batch200", c2st=score)
    assert 0.42 <= score <= 0.6, score


@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_sampling_direction_matches_oracle_and_throughput_kernel(cfg):
    """Small sampling calls run on the cooperative inverse kernel (one-m-tile instantiation of nsf_coopw_fwd_kernel with
    the packed LU inverses): against the oracle, against the throughput inverse kernel, and as the inverse of log_prob's
    transform, ragged row counts included."""
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    g = torch.Generator().manual_seed(5)
    for n in (1, 17, 600, 5000):
        noise = torch.randn(n, cfg["D"], generator=g)
        x = x_d[:n] if n <= x_d.shape[0] else x_d[torch.randint(0, x_d.shape[0], (n,), generator=g)]
        with torch.no_grad():
            ref, ref_ld = oracle.sample_from_noise(noise, x)
        with family("cooperative"):
            got, got_ld = est.sample_from_noise(noise.cuda(), x.cuda(), with_logabsdet=True)
            back = est.inverse_transform(got, x.cuda())
        with family("throughput"):
            try:
                thr, thr_ld = est.sample_from_noise(noise.cuda(), x.cuda(), with_logabsdet=True)
            except RuntimeError:
                thr = None
        scale = max(1.0, ref.abs().max().item())
        e = (got.cpu() - ref).abs().max().item()
        e_ld = (got_ld.cpu() - ref_ld).abs().max().item()
        record("coop_sample", _ids(cfg) + f" | n={n}", max_abs_coop_vs_oracle32=e, max_abs_logabsdet=e_ld,
               max_abs_ref=ref.abs().max().item())
        assert e <= 2e-5 * scale and e_ld <= 2e-5 * max(1.0, ref_ld.abs().max().item()) + 1e-5, (n, e, e_ld)
        assert (back.cpu() - noise).abs().max().item() <= 2e-4 * max(1.0, noise.abs().max().item()), n
        if thr is not None:
            assert (got - thr).abs().max().item() <= 4e-5 * scale, n
            assert (got_ld - thr_ld).abs().max().item() <= 4e-5 * max(1.0, ref_ld.abs().max().item()) + 2e-5, n
