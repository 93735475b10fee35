"""Oracle comparisons AT THE BENCHMARKED SIZE (BASELINE configs[1]: theta-dim 10, x-dim 10, batch 65 536; FMPE
theta-dim = x-dim = 50, batch 65 536): log_prob (paired x and one broadcast x_o), sample for given noise, the flat
training gradient of the fused step (persistent backward kernel with many tiles per workgroup) against autograd
through the CPU oracle, and the FMPE loss / gradient against its pinned oracle.  The CPU oracle runs 65 536 rows
in seconds; its autograd pass is chunked (16 384 rows) to bound host memory -- gradients of a mean are additive.

Every test records its measured distances in the parity artifact (tests/parity_log.py -> profiles/parity_rN.json).
"""

import pytest
import torch

from tests.helpers import matched_pair, row_parity
from tests.parity_log import record

pytestmark = pytest.mark.gpu

N = 65536
CHUNK = 16384


def _bench_data(seed=0):
    """bench.py's synthetic linear-Gaussian batch."""
    g = torch.Generator().manual_seed(seed)
    theta = torch.randn(N, 10, generator=g) * (0.1**0.5)
    x = theta + (0.1**0.5) * torch.randn(N, 10, generator=g)
    return theta, x


def _oracle_log_prob(oracle, theta, x, double=False):
    out = []
    with torch.no_grad():
        if double:
            oracle.double()
        for i in range(0, theta.shape[0], CHUNK):
            th, xx = theta[i : i + CHUNK], x[i : i + CHUNK] if x.shape[0] == theta.shape[0] else x
            if double:
                th, xx = th.double(), xx.double()
            if xx.shape[0] != th.shape[0]:
                xx = xx.expand(th.shape[0], -1)
            out.append(oracle.log_prob(th, xx)[0])
        if double:
            oracle.float()
    return torch.cat(out)


@pytest.mark.parametrize("mode", ["paired_x", "broadcast_x_o"])
def test_log_prob_65536_matches_oracle(mode):
    oracle, est, _, _ = matched_pair(D=10, C=10)
    theta, x = _bench_data()
    xx = x if mode == "paired_x" else x[:1]
    ref = _oracle_log_prob(oracle, theta, xx)
    ref64 = _oracle_log_prob(oracle, theta, xx, double=True)
    if mode == "paired_x":
        got = est.log_prob(theta.cuda(), xx.cuda())[0].cpu()
    else:
        got = est.log_prob(theta.cuda().unsqueeze(1), xx.cuda())[:, 0].cpu()
    e_o = (got - ref).abs().max().item()
    e_hip = (got.double() - ref64).abs().max().item()
    e_ref = (ref.double() - ref64).abs().max().item()
    record("log_prob_65536", mode, max_abs_hip_vs_oracle32=e_o, max_abs_hip_vs_f64=e_hip,
           max_abs_oracle32_vs_f64=e_ref, max_abs_ref=ref.abs().max().item(), rows=N,
           frac_rows_within_1e5_of_oracle32=((got - ref).abs() <= 1e-5).float().mean().item(),
           frac_rows_within_1e5_of_f64=((got.double() - ref64).abs() <= 1e-5).float().mean().item(),
           frac_oracle32_rows_within_1e5_of_f64=((ref.double() - ref64).abs() <= 1e-5).float().mean().item())
    print(f"{mode}: |hip-o32|={e_o:.3e} |hip-f64|={e_hip:.3e} |o32-f64|={e_ref:.3e} max|ref|={ref.abs().max():.1f}")
    assert torch.isfinite(got).all()
    assert e_o <= 1e-5 + 1e-5 * ref.abs().max().item()
    assert e_hip <= 2.0 * e_ref + 1e-5
    # row by row: |d_i| <= 1e-5 (1 + |ref_i|) on all but 1 % of the 65 536 rows, no row beyond 4 x its bound
    rp, rp64, ro64 = row_parity(got, ref), row_parity(got, ref64), row_parity(ref, ref64)
    record("log_prob_65536_rows", mode, **{f"hip_vs_o32.{k}": v for k, v in rp.items()},
           **{f"hip_vs_f64.{k}": v for k, v in rp64.items()}, **{f"o32_vs_f64.{k}": v for k, v in ro64.items()})
    print(f"{mode} per-row: hip vs o32 worst {rp['worst_scaled']:.2f} x bound, beyond {rp['exceed_frac']:.3%}; hip vs "
          f"f64 worst {rp64['worst_scaled']:.2f}, beyond {rp64['exceed_frac']:.3%}; o32 vs f64 worst "
          f"{ro64['worst_scaled']:.2f}, beyond {ro64['exceed_frac']:.3%}")
    assert rp["exceed_frac"] <= 0.01 and rp["worst_scaled"] <= 4.0, rp
    assert rp64["exceed_frac"] <= 0.01 and rp64["worst_scaled"] <= 4.0, rp64
    # north_star's "within 1e-5" read ABSOLUTELY, against the fp64 evaluation: >= 99 % of the 65 536 rows (round 5: the
    # spline's selected bin is re-derived with the softmax denominator / prefix sum in fp64, csrc/nsf_device.h
    # precise_bin; measured 0.2 % beyond -- the eager fp32 oracle itself: 9 %, which is also what bounds hip vs o32)
    assert rp64["abs_exceed_frac"] <= 0.01, rp64
    assert rp64["abs_exceed_frac"] <= 0.5 * ro64["abs_exceed_frac"] + 1e-3, (rp64, ro64)   # strictly closer than eager fp32


def test_sample_from_noise_65536_matches_oracle():
    oracle, est, _, _ = matched_pair(D=10, C=10)
    _, x = _bench_data()
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(N, 10, generator=g)
    outs, outs64 = [], []
    with torch.no_grad():
        for i in range(0, N, CHUNK):
            outs.append(oracle.sample_from_noise(noise[i : i + CHUNK], x[i : i + CHUNK])[0])
        oracle.double()
        for i in range(0, N, CHUNK):
            outs64.append(oracle.sample_from_noise(noise[i : i + CHUNK].double(), x[i : i + CHUNK].double())[0])
        oracle.float()
    ref, ref64 = torch.cat(outs), torch.cat(outs64)
    got = est.sample_from_noise(noise.cuda(), x.cuda()).cpu()
    e_o = (got - ref).abs().max().item()
    e_hip = (got.double() - ref64).abs().max().item()
    e_ref = (ref.double() - ref64).abs().max().item()
    frac = ((got - ref).abs() <= 1e-5).float().mean().item()
    record("sample_65536", "D10-C10", max_abs_hip_vs_oracle32=e_o, max_abs_hip_vs_f64=e_hip,
           max_abs_oracle32_vs_f64=e_ref, max_abs_ref=ref.abs().max().item(), rows=N,
           frac_entries_within_1e5_of_oracle32=frac)
    print(f"sample: |hip-o32|={e_o:.3e} |hip-f64|={e_hip:.3e} |o32-f64|={e_ref:.3e} within 1e-5: {frac:.5%}")
    assert e_hip <= 2.0 * e_ref + 1e-5
    assert frac > 0.999


KNOT_ULPS = 2.0             # fp32 spacings at the tail bound within which a row counts as knot-straddling

GRAD_CONFIGS = {
    "D10-C10": dict(D=10, C=10),                                     # BASELINE configs[1]: wave-specialised backward
    "generic-D20-C12-T3": dict(D=20, C=12, num_transforms=3),        # generic pass (nsf_gtrain_kernel.h + dW GEMMs)
}


@pytest.mark.parametrize("name", list(GRAD_CONFIGS))
def test_training_gradient_65536_matches_autograd(name):
    """The fused step's flat gradient at the benchmarked batch (256 persistent workgroups x 4 tiles each; for the
    generic pass: 128 split-K chunks per linear).

    At this size (65 536 rows x 25 spline evaluations) a handful of spline inputs land within one fp32 ulp of a
    knot.  The RQ spline is C1: log p is continuous there, but d log p / d(parameters, theta) is two-valued (the
    second derivative jumps at a knot), and the kernels' knot positions differ from the eager oracle's by an ulp
    (fused multiply-adds), so such a row can take the neighbouring bin: its gradient contribution (weight 1/N) then
    differs by O(1) x 1/N.  The test finds those rows through the per-row d loss / d theta (every other row agrees
    closely), bounds their number, and holds the flat gradient of all remaining rows to the fp64 oracle."""
    from sbi_amd.neural_nets.estimators.nsf_flow import loss_fwd_bwd, train_workspace
    from tests.test_nsf_train_gpu import oracle_flat_grad

    cfg = GRAD_CONFIGS[name]
    oracle, est, _, _ = matched_pair(**cfg)
    if cfg["D"] == 10 and cfg["C"] == 10:
        theta, x = _bench_data(seed=2)
    else:
        from tests.helpers import linear_gaussian_data

        theta, x = linear_gaussian_data(N, cfg["D"], cfg["C"], seed=2)

    def oracle_pass(double, keep=None):
        """(per-row loss, flat param grad of sum_n keep_n loss_n / N, per-row d loss_n / d theta_n)"""
        oracle.zero_grad()
        dt = torch.float64 if double else torch.float32
        oracle.double() if double else oracle.float()
        losses, gth = [], []
        for i in range(0, N, CHUNK):
            th = theta[i : i + CHUNK].to(dt).requires_grad_(True)
            l = oracle.loss(th, x[i : i + CHUNK].to(dt))
            w = torch.ones(l.shape[0], dtype=dt) if keep is None else keep[i : i + CHUNK].to(dt)
            ((l * w).sum() / N).backward()
            losses.append(l.detach())
            gth.append(th.grad * N)        # rows are independent: row n of th.grad is d loss_n / d theta_n / N
        named = dict(oracle.named_parameters())
        flat = torch.zeros(est.net.flat_params.numel(), dtype=dt)
        for key, off, n_, _ in est.net._slices():
            flat[off : off + n_] = named["net." + key].grad.reshape(-1)
        oracle.float()
        return torch.cat(losses), flat, torch.cat(gth)

    def hip_pass(keep=None):
        grad = torch.empty_like(est.net.flat_params.data)
        ws = train_workspace(est.net, N, "cuda")
        ws.fill_(float("nan"))
        rw = None if keep is None else (keep / N).cuda().contiguous()
        losses, gth = loss_fwd_bwd(est.net, theta.cuda(), x.cuda(), rw, 1.0 / N, grad, want_grad_theta=True,
                                   workspace=ws)
        torch.cuda.synchronize()
        return losses.cpu(), grad.cpu(), gth.cpu() * N

    loss32, g32, _ = oracle_pass(False)
    loss64, g64, gth64 = oracle_pass(True)
    loss_h, g_h, gth_h = hip_pass()
    assert torch.isfinite(g_h).all() and torch.isfinite(gth_h).all()
    scale = g64.abs().max().item()
    e_l = (loss_h - loss32).abs().max().item()
    assert e_l <= 1e-5 + 1e-5 * loss32.abs().max().item()
    e_all = (g_h.double() - g64).abs().max().item() / scale
    e_o64 = (g32.double() - g64).abs().max().item() / scale
    # rows whose d loss / d theta disagrees with the fp64 oracle: the knot-straddling rows
    row_err = (gth_h.double() - gth64).abs().max(dim=1).values / gth64.abs().max().item()
    outliers = (row_err > 1e-3).nonzero().flatten()
    typical = row_err[row_err <= 1e-3].max().item()
    print(f"grad 65536, all rows: hip vs f64 {e_all:.3e}, o32 vs f64 {e_o64:.3e}; rows off in d loss/d theta: "
          f"{outliers.tolist()} (err {row_err[outliers].tolist()}), every other row within {typical:.3e}")
    assert outliers.numel() <= 8, "more knot-straddling rows than one-ulp knot differences can explain"
    # VERIFY that every row set aside really straddles a knot: in float64 through the oracle, at least one of its
    # T x d_tr spline inputs must lie within KNOT_ULPS float32 spacings (at the tail bound: 2.4e-7) of an interior
    # knot -- the distance within which two correct fp32 evaluations of softmax + cumsum can disagree about the bin.
    # A row that is merely WRONG (far from every knot) fails here instead of being excused.
    from tests.helpers import spline_knot_distances

    near = torch.zeros(0)
    if outliers.numel():
        near = spline_knot_distances(oracle, theta[outliers], x[outliers]).min(dim=1).values
        print(f"excluded rows: distance of the closest spline input to a knot, in fp32 spacings at B: {near.tolist()}")
        assert (near <= KNOT_ULPS).all(), \
            f"row(s) {outliers[near > KNOT_ULPS].tolist()} disagree with fp64 but sit {near.tolist()} spacings from a knot"
    ctrl = spline_knot_distances(oracle, theta[:4096], x[:4096]).min(dim=1).values     # what ordinary rows look like
    keep = torch.ones(N)
    keep[outliers] = 0.0
    _, g64k, _ = oracle_pass(True, keep)
    _, g32k, _ = oracle_pass(False, keep)
    _, g_hk, _ = hip_pass(keep)
    e_keep = (g_hk.double() - g64k).abs().max().item() / scale
    # per parameter block, relative to the block's own largest entry (floored at 1e-3 of the global maximum): the
    # eager fp32 oracle is held to the same yardstick, so a block whose entries are small next to their own
    # rounding noise (near-zero-gradient biases) shows up as such instead of as a kernel error
    worst_block, worst_name, worst_o32, worst_o32_name, worst_mag = 0.0, "", 0.0, "", 0.0
    for key, off, cnt, _ in est.net._slices():
        b = g64k[off : off + cnt]
        den = max(b.abs().max().item(), 1e-3 * scale)
        eh = (g_hk[off : off + cnt].double() - b).abs().max().item() / den
        eo = (g32k[off : off + cnt].double() - b).abs().max().item() / den
        if eh > worst_block:
            worst_block, worst_name, worst_mag = eh, key, b.abs().max().item() / scale
        if eo > worst_o32:
            worst_o32, worst_o32_name = eo, key
    record("train_grad_65536", name, rows=N, max_abs_grad_ref=scale, max_abs_loss_err_vs_oracle32=e_l,
           rel_grad_err_hip_vs_f64_all_rows=e_all, rel_grad_err_oracle32_vs_f64_all_rows=e_o64,
           knot_straddling_rows=int(outliers.numel()), rel_grad_err_hip_vs_f64_without_those_rows=e_keep,
           worst_block_rel_err_without_those_rows=worst_block, worst_block=worst_name,
           worst_block_max_over_global_max=worst_mag, worst_block_rel_err_oracle32=worst_o32,
           worst_block_oracle32=worst_o32_name, max_rel_row_grad_theta_err_other_rows=typical,
           excluded_rows_min_knot_distance_fp32_spacings=[float(v) for v in near.tolist()],
           control_rows_min_knot_distance_fp32_spacings_median=ctrl.median().item(),
           control_rows_min_knot_distance_fp32_spacings_min=ctrl.min().item(), knot_ulps_bound=KNOT_ULPS)
    print(f"without those rows: hip vs f64 {e_keep:.3e}, worst block {worst_name} {worst_block:.3e} (block max = "
          f"{worst_mag:.2e} of the global max); fp32 oracle's worst block {worst_o32_name} {worst_o32:.3e}")
    assert e_keep <= 5e-5, f"flat gradient off by {e_keep} of max|grad| on rows away from knots"
    # Per block, measured against the GLOBAL largest gradient entry the kernels must be as good on every block as
    # they are overall (5e-5 of max|grad|).  Measured against the block's OWN largest entry the bar is 5e-4: the worst
    # block (r3: blocks.0.linear_layers.0.weight of the last transform, 3.5e-4) is one whose entries are 0.4 % of the
    # global maximum -- sums of 65 536 row contributions that cancel to 1/250 of their typical size, so its absolute
    # error (1.4e-6 of max|grad|) is BELOW the flat gradient's worst entry (4.1e-6) while its relative error reads
    # large; the eager fp32 oracle shows the same pattern on the same block (3.3e-5: it accumulates the batch in
    # 16 384-row chunks of a blocked GEMM, the kernels in 256 slabs of 256 rows).  tools/diag/knot_rows.py lists the
    # per-row errors: only the two excluded rows sit on a knot, every other row is within 1.7e-4 of its own d loss /
    # d theta and far from any knot -- ordinary fp32 conditioning, not bin flips.
    assert worst_block * worst_mag <= 5e-5, (worst_name, worst_block, worst_mag)
    assert worst_block <= 5e-4


def test_fmpe_loss_and_gradient_65536_match_pinned_oracle():
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import loss_fwd_bwd
    from tests.test_fmpe_gpu import flat_grad_of, make_pair

    oracle, est, theta, x, times, noise = make_pair(50, 50, n=N)
    ref_losses = []
    for i in range(0, N, CHUNK):
        s = slice(i, i + CHUNK)
        l = oracle.loss(theta[s], x[s], times[s], noise[s])
        (l.sum() / N).backward()
        ref_losses.append(l.detach())
    ref_l = torch.cat(ref_losses)
    gref = flat_grad_of(oracle, est)
    grad = torch.empty_like(est.net.flat_params.data)
    got_l = loss_fwd_bwd(est.net, theta.cuda(), x.cuda(), times.cuda(), noise.cuda(), None, 1.0 / N, grad).cpu()
    e_l = (got_l - ref_l).abs().max().item() / ref_l.abs().max().item()
    scale = gref.abs().max().item()
    e_g = (grad.cpu() - gref).abs().max().item() / scale
    record("fmpe_65536", "D50-C50", rel_loss_err=e_l, rel_grad_err=e_g, max_abs_grad_ref=scale, rows=N)
    print(f"fmpe 65536: rel loss err {e_l:.3e} rel grad err {e_g:.3e}")
    assert e_l <= 2e-5
    assert e_g <= 3e-4
