"""Oracle comparisons AT THE BENCHMARKED SIZE (BASELINE configs[1]: theta-dim 10, x-dim 10, batch 65 536; FMPE
theta-dim = x-dim = 50, batch 65 536): log_prob (paired x and one broadcast x_o), sample for given noise, the flat
training gradient of the fused step (persistent backward kernel with many tiles per workgroup) against autograd
through the CPU oracle, and the FMPE loss / gradient against its pinned oracle.  The CPU oracle runs 65 536 rows
in seconds; its autograd pass is chunked (16 384 rows) to bound host memory -- gradients of a mean are additive.

Every test records its measured distances in the parity artifact (tests/parity_log.py -> profiles/parity_rN.json).
"""

import pytest
import torch

from tests.helpers import matched_pair
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


def test_training_gradient_65536_matches_autograd():
    """The fused step's flat gradient at the benchmarked batch (256 persistent workgroups x 4 tiles each)."""
    from sbi_amd.inference.trainers.fused import FusedTrainStep
    from tests.test_nsf_train_gpu import oracle_flat_grad

    oracle, est, _, _ = matched_pair(D=10, C=10)
    theta, x = _bench_data(seed=2)
    oracle.zero_grad()
    losses_ref = []
    for i in range(0, N, CHUNK):        # gradient of the batch mean, accumulated over chunks
        l = oracle.loss(theta[i : i + CHUNK], x[i : i + CHUNK])
        (l.sum() / N).backward()
        losses_ref.append(l.detach())
    loss_ref = torch.cat(losses_ref)
    gref = oracle_flat_grad(oracle, est)
    # the same gradient in fp64: a 65 536-term fp32 sum has its own round-off, in the oracle as in the kernels
    oracle.zero_grad()
    oracle.double()
    for i in range(0, N, CHUNK):
        (oracle.loss(theta[i : i + CHUNK].double(), x[i : i + CHUNK].double()).sum() / N).backward()
    named = dict(oracle.named_parameters())
    gref64 = torch.zeros(est.net.flat_params.numel(), dtype=torch.float64)
    for key, off, n_, _ in est.net._slices():
        gref64[off : off + n_] = named["net." + key].grad.reshape(-1)
    oracle.float()
    stepper = FusedTrainStep(est, distributed=False)
    stepper._workspace(N).fill_(float("nan"))
    losses = stepper.loss_and_grad(theta.cuda(), x.cuda())
    torch.cuda.synchronize()
    got = stepper.grad.cpu()
    scale = gref.abs().max().item()
    err = (got - gref).abs().max().item()
    e_l = (losses.cpu() - loss_ref).abs().max().item()
    worst_block = 0.0
    for key, off, cnt, _ in est.net._slices():
        a, b = got[off : off + cnt], gref[off : off + cnt]
        worst_block = max(worst_block, (a - b).abs().max().item() / max(b.abs().max().item(), 1e-3 * scale))
    e_hip64 = (got.double() - gref64).abs().max().item() / scale
    e_o64 = (gref.double() - gref64).abs().max().item() / scale
    record("train_grad_65536", "D10-C10", max_abs_grad_err_vs_oracle32=err, max_abs_grad_ref=scale,
           rel_grad_err_vs_oracle32=err / scale, rel_grad_err_hip_vs_f64=e_hip64, rel_grad_err_oracle32_vs_f64=e_o64,
           worst_block_rel_err_vs_oracle32=worst_block, max_abs_loss_err=e_l,
           max_abs_loss_ref=loss_ref.abs().max().item(), rows=N)
    print(f"grad 65536: rel vs o32 {err / scale:.3e} (worst block {worst_block:.3e}), hip vs f64 {e_hip64:.3e}, "
          f"o32 vs f64 {e_o64:.3e}, loss err {e_l:.3e}")
    assert torch.isfinite(got).all()
    assert e_l <= 1e-5 + 1e-5 * loss_ref.abs().max().item()
    # bar: within 2e-4 of max|grad| of the fp64 gradient, or no worse than twice the fp32 oracle's own distance
    assert e_hip64 <= max(2e-4, 2.0 * e_o64), f"hip {e_hip64} vs oracle32 {e_o64} (both against fp64)"


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
