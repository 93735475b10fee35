"""The broadcast-x specialisation of the throughput kernels (`nsf_flow_kernel<..., BX = true>`, csrc/nsf_device.h
`bx_fold_context`): with ONE condition row for the whole launch -- every `DirectPosterior.sample / log_prob`, MCMC
potential and rejection call (sbi/inference/posteriors/direct_posterior.py:358-364, the reference copies x_o N times,
sbi/neural_nets/estimators/base.py:142-198) -- the context-only terms of each transform's conditioner are computed
once per workgroup instead of once per row.  Held to (i) the oracle and (ii) the SAME kernels called with x_o expanded
to one row per theta row (the per-row path), on shapes that exercise the dynamic plan: odd dims, x-dim > 16, one
block, 64 hidden units, every bin count, no z-scoring."""
import pytest
import torch

from tests.helpers import matched_pair, make_inputs, row_parity

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(D=10, C=10),
    dict(D=2, C=2),
    dict(D=4, C=7),
    dict(D=3, C=5, hidden_features=32, num_transforms=3, num_bins=8, num_blocks=1),
    dict(D=5, C=3, hidden_features=64, num_transforms=4, num_bins=5),
    dict(D=10, C=10, z_score_theta="none", z_score_x="none"),
    dict(D=6, C=12, num_bins=16, num_transforms=2),
    dict(D=7, C=4, num_bins=4, hidden_features=20, tail_bound=5.0),
    dict(D=9, C=40),                                    # x-dim > 16: the context rows live in LDS on the per-row path
    dict(D=12, C=24, num_blocks=1),
]


def _ids(c):
    return "-".join(f"{k}{v}" for k, v in c.items())


@pytest.fixture(autouse=True)
def throughput_family():
    """The specialisation lives in the throughput kernels: switch the cooperative small-batch family off so that every
    size below reaches them (60 rows: one wave per workgroup ... 20 000 rows: eight)."""
    from sbi_amd import _lib

    prev = _lib.load().sbi_amd_nsf_set_coop_max_rows(0)
    yield
    _lib.load().sbi_amd_nsf_set_coop_max_rows(prev)


@pytest.mark.parametrize("n", [60, 1000, 20000])
@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_log_prob_one_x_o(cfg, n):
    oracle, est, _, x_d = matched_pair(**cfg)
    theta, _ = make_inputs(n, cfg["D"], cfg["C"])
    theta[::7] /= 6.0      # (keep the rows in distribution: make_inputs pushes every 7th into the tails)
    x_o = x_d[3:4]
    with torch.no_grad():
        ref = oracle.log_prob(theta.unsqueeze(1), x_o)[:, 0]
    got = est.log_prob(theta.cuda().unsqueeze(1), x_o.cuda())[:, 0].cpu()                  # x_rows == 1: BX
    per_row = est.log_prob(theta.cuda(), x_o.cuda().expand(n, -1).contiguous())[0].cpu()   # x_rows == n: per-row path
    assert torch.isfinite(got).all()
    rp, rs = row_parity(got, ref), row_parity(got, per_row)
    print(f"{_ids(cfg)} n={n}: vs oracle worst {rp['worst_scaled']:.2f} x bound ({rp['exceed_frac']:.3%} beyond), vs the "
          f"per-row kernels worst {rs['worst_scaled']:.2f} x bound, max |d| {rs['max_abs']:.2e}")
    # (one row of 60 is already 1.7 %; the eager fp32 oracle itself misses its fp64 evaluation by more than the bound on
    #  ~9 % of rows, DESIGN.md section 2, so at n = 60 two rows may land beyond it: worst 1.2 x the bound in round 6)
    few = max(0.01, 2.5 / n)
    assert rp["exceed_frac"] <= few and rp["worst_scaled"] <= 4.0, rp
    assert rs["exceed_frac"] <= few and rs["worst_scaled"] <= 4.0, rs


@pytest.mark.parametrize("n", [60, 20000])
@pytest.mark.parametrize("cfg", CONFIGS, ids=_ids)
def test_sample_one_x_o(cfg, n):
    oracle, est, _, x_d = matched_pair(**cfg)
    noise = torch.randn(n, cfg["D"], generator=torch.Generator().manual_seed(5))
    x_o = x_d[5:6]
    with torch.no_grad():
        ref, ref_ld = oracle.sample_from_noise(noise, x_o)
    got, got_ld = est.sample_from_noise(noise.cuda(), x_o.cuda(), with_logabsdet=True)
    per_row, per_row_ld = est.sample_from_noise(noise.cuda(), x_o.cuda().expand(n, -1).contiguous(), with_logabsdet=True)
    e_o = (got.cpu() - ref).abs()
    frac = (e_o <= 1e-5 * (1 + ref.abs())).float().mean().item()
    e_s = (got - per_row).abs().max().item()
    e_ld = (got_ld.cpu() - ref_ld).abs().max().item()
    print(f"{_ids(cfg)} n={n}: theta within 1e-5 of the oracle on {frac:.4%}, max {e_o.max().item():.2e}; vs per-row "
          f"kernels {e_s:.2e}; logabsdet vs oracle {e_ld:.2e}")
    assert frac >= 0.999 and e_o.max().item() <= 1e-4
    assert e_s <= 5e-5
    assert e_ld <= 1e-5 * (1 + ref_ld.abs().max().item()) * 4


def test_direct_posterior_calls_take_the_specialisation():
    """`DirectPosterior.log_prob / sample` hand the estimator ONE x_o row (no expand): the C ABI sees x_rows == 1."""
    from sbi_amd.inference import DirectPosterior
    from sbi_amd.utils.torchutils import BoxUniform

    oracle, est, _, x_d = matched_pair(D=10, C=10)
    prior = BoxUniform(-3.0 * torch.ones(10), 3.0 * torch.ones(10), device="cuda")
    post = DirectPosterior(est, prior, device="cuda").set_default_x(x_d[:1])
    torch.manual_seed(0)
    s = post.sample((30000,), show_progress_bars=False)
    lp = post.log_prob(s, norm_posterior=False).cpu()
    with torch.no_grad():
        ref = torch.cat([oracle.log_prob(s[i : i + 10000].cpu().unsqueeze(1), x_d[:1])[:, 0] for i in range(0, 30000, 10000)])
    rp = row_parity(lp, ref)
    assert rp["exceed_frac"] <= 0.01 and rp["worst_scaled"] <= 4.0, rp
