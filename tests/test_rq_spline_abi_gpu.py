"""`sbi_amd_rq_spline`: the coupling transform's rational-quadratic spline through its own C-ABI entry point
(include/sbi_amd_nsf.h; SURVEY 8b `spline_coupling_fwd / inv`) against the oracle's restatement of nflows'
`unconstrained_rational_quadratic_spline` -- random and saturated parameters, inputs inside, on and beyond the tail
bound, every bin count, both directions, value and log-determinant, and the round trip."""
import pytest
import torch

from oracle.nsf_oracle import unconstrained_rational_quadratic_spline
from sbi_amd import _lib

pytestmark = pytest.mark.gpu


def hip_spline(params, inputs, K, inverse, B=3.0, scale=1.0):
    lib = _lib.load()
    n = inputs.shape[0]
    p, x = params.cuda().contiguous(), inputs.cuda().contiguous()
    out, ld = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    with torch.cuda.device(0):
        rc = lib.sbi_amd_rq_spline(K, int(inverse), B, 1e-3, 1e-3, 1e-3, scale, _lib.ptr(p), _lib.ptr(x), n, _lib.ptr(out),
                                   _lib.ptr(ld), _lib.current_stream(torch.device("cuda", 0)))
    _lib.check(rc, "rq_spline")
    return out.cpu(), ld.cpu()


def oracle_spline(params, inputs, K, inverse, B=3.0, scale=1.0, dtype=torch.float64):
    p, x = params.to(dtype), inputs.to(dtype)
    uw, uh, ud = p[:, :K] * scale, p[:, K : 2 * K] * scale, p[:, 2 * K :]
    return unconstrained_rational_quadratic_spline(x, uw, uh, ud, inverse=inverse, tail_bound=B)


@pytest.mark.parametrize("K", [4, 5, 8, 10, 16])
@pytest.mark.parametrize("n", [1, 33, 4097])
def test_spline_value_and_logdet_both_directions(K, n):
    g = torch.Generator().manual_seed(K * 100 + n)
    params = torch.randn(n, 3 * K - 1, generator=g) * 2.0
    x = torch.randn(n, generator=g) * 1.8
    if n > 8:
        x[0], x[1], x[2], x[3] = 3.0, -3.0, 3.5, -7.0          # on the bound and in the linear tails
        params[4] = 0.0                                          # uniform bins
    y, ld = hip_spline(params, x, K, False)
    y64, ld64 = oracle_spline(params, x, K, False)
    y32, ld32 = oracle_spline(params, x, K, False, dtype=torch.float32)
    ey, el = (y.double() - y64).abs().max().item(), (ld.double() - ld64).abs().max().item()
    oy, ol = (y32.double() - y64).abs().max().item(), (ld32.double() - ld64).abs().max().item()
    print(f"K={K} n={n} forward: |y - f64| {ey:.2e} (eager fp32 {oy:.2e}); |logdet - f64| {el:.2e} (eager fp32 {ol:.2e})")
    assert ey <= 2e-6 + 2 * oy and el <= 2e-6 + 2 * ol
    # inverse direction.  With raw logits of this spread a bin's slope ranges over 1e-3 ... 1e3 and either direction
    # amplifies an fp32 ulp by that factor somewhere (tests/test_spline_adversarial_gpu.py holds that regime to careful
    # yardsticks): here only finiteness.  The strict comparison runs at the operating point of sbi's couplings -- logits
    # divided by sqrt(hidden_features) -- where slopes stay within ~[0.1, 10].
    xw, lw = hip_spline(params, y, K, True)
    assert torch.isfinite(xw).all() and torch.isfinite(lw).all()
    scale = 50.0 ** -0.5
    ys, lds = hip_spline(params, x, K, False, scale=scale)
    xb, ldb = hip_spline(params, ys, K, True, scale=scale)
    xb64, ldb64 = oracle_spline(params, ys, K, True, scale=scale)
    xb32, ldb32 = oracle_spline(params, ys, K, True, scale=scale, dtype=torch.float32)
    ex, ox = (xb.double() - xb64).abs().max().item(), (xb32.double() - xb64).abs().max().item()
    elb, olb = (ldb.double() - ldb64).abs().max().item(), (ldb32.double() - ldb64).abs().max().item()
    print(f"K={K} n={n} inverse (logits / sqrt(50)): |x - f64| {ex:.2e} (eager fp32 {ox:.2e}); |logdet - f64| {elb:.2e} "
          f"(eager fp32 {olb:.2e}); round trip {(xb - x).abs().max().item():.2e}")
    assert ex <= 5e-6 + 3 * ox and elb <= 1e-5 + 3 * olb
    assert (xb - x).abs().max().item() <= 3e-5      # (the two log-determinants are each held to fp64 above)


def test_spline_logit_scale_and_refusals():
    K, n = 10, 257
    g = torch.Generator().manual_seed(1)
    params = torch.randn(n, 3 * K - 1, generator=g) * 5.0
    x = torch.randn(n, generator=g)
    scale = 50.0 ** -0.5              # sbi's couplings: logits / sqrt(hidden_features)
    y, ld = hip_spline(params, x, K, False, scale=scale)
    y64, ld64 = oracle_spline(params, x, K, False, scale=scale)
    y32, ld32 = oracle_spline(params, x, K, False, scale=scale, dtype=torch.float32)
    assert (y.double() - y64).abs().max().item() <= 2e-6 + 2 * (y32.double() - y64).abs().max().item()
    assert (ld.double() - ld64).abs().max().item() <= 2e-6 + 2 * (ld32.double() - ld64).abs().max().item()
    lib = _lib.load()
    buf = torch.zeros(64, device="cuda")
    assert lib.sbi_amd_rq_spline(7, 0, 3.0, 1e-3, 1e-3, 1e-3, 1.0, _lib.ptr(buf), _lib.ptr(buf), 2, _lib.ptr(buf), None,
                                 None) == _lib.E_UNSUPPORTED
    assert lib.sbi_amd_rq_spline(10, 0, -1.0, 1e-3, 1e-3, 1e-3, 1.0, _lib.ptr(buf), _lib.ptr(buf), 2, _lib.ptr(buf), None,
                                 None) < 0
    assert lib.sbi_amd_rq_spline(10, 0, 3.0, 1e-3, 1e-3, 1e-3, 1.0, None, None, 0, None, None, None) == 0     # n = 0: no-op
