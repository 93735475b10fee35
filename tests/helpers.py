"""Shared test helpers: matched (oracle, HIP estimator) pairs on seeded data."""

import torch

from oracle.nsf_oracle import NSFOracle
from sbi_amd.neural_nets.estimators.nsf_flow import loss_fwd_bwd, train_workspace
from sbi_amd.neural_nets.net_builders.flow import build_nsf


def linear_gaussian_data(n, D, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    theta = torch.randn(n, D, generator=g) * (0.1**0.5)
    A = torch.randn(D, C, generator=g) / D**0.5 if C != D else torch.eye(D)
    x = theta @ A + (0.1**0.5) * torch.randn(n, C, generator=g)
    return theta, x


def matched_pair(D=10, C=10, n=1000, perturb=0.05, seed=1, device="cuda", **kw):
    """Oracle + HIP estimator with identical (perturbed) weights and z-score stats."""
    theta, x = linear_gaussian_data(n, D, C)
    okw = dict(hidden_features=kw.get("hidden_features", 50), num_transforms=kw.get("num_transforms", 5),
               num_bins=kw.get("num_bins", 10), num_blocks=kw.get("num_blocks", 2),
               tail_bound=kw.get("tail_bound", 3.0))
    if "hidden_layers_spline_context" in kw:
        okw["hidden_layers_spline_context"] = kw["hidden_layers_spline_context"]
    zt = kw.get("z_score_theta", "independent")
    zx = kw.get("z_score_x", "independent")
    torch.manual_seed(seed)
    oracle = NSFOracle(theta, x, z_score_theta=zt, z_score_x=zx, **okw)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for p in oracle.parameters():
            p.add_(perturb * torch.randn(p.shape, generator=g))
    est = build_nsf(theta, x, z_score_x=zt, z_score_y=zx, **okw)
    est.net.load_nflows_state_dict(oracle.state_dict())
    if device is not None:
        est = est.to(device)
    return oracle, est, theta, x


from sbi_amd.utils.parity import row_parity  # noqa: E402,F401  (kept importable from here)


def make_inputs(n, D, C, seed=3, spread=1.0):
    """Rows that exercise the spline interior, the linear tails and exact +-bound hits."""
    g = torch.Generator().manual_seed(seed)
    theta = torch.randn(n, D, generator=g) * (0.1**0.5) * spread
    x = torch.randn(n, C, generator=g) * (0.2**0.5) * spread
    theta[::7] *= 6.0   # push some rows far into the tails
    return theta, x


def spline_knot_distances(oracle, theta, x, include_bounds=True):
    """For every row: the distance of each forward spline evaluation's input to its nearest INTERIOR knot, computed
    in float64 through the oracle, in units of the float32 spacing at the tail bound (np.spacing(float32(B)): the
    resolution at which an fp32 implementation can place a knot of a spline on [-B, B]).  Returns
    (rows, num_transforms * d_tr) -- 25 evaluations per row at theta-dim 10.  The tail bounds +-B count as knots
    (`include_bounds`; otherwise inputs in the tails get +inf).

    Used by the full-size gradient test to VERIFY that the rows it sets aside are knot-straddling rows (the RQ spline
    is C1: at a knot log p is continuous but its parameter / input gradient is two-valued)."""
    import numpy as np
    import torch.nn.functional as F

    import oracle.nsf_oracle as mod

    captured = []
    real = mod.unconstrained_rational_quadratic_spline

    def spy(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse=False,
            tail_bound=1.0, min_bin_width=mod.DEFAULT_MIN_BIN_WIDTH, **kw):
        K = unnormalized_widths.shape[-1]
        w = min_bin_width + (1 - min_bin_width * K) * F.softmax(unnormalized_widths, dim=-1)
        knots = 2 * tail_bound * torch.cumsum(w, dim=-1)[..., :-1] - tail_bound          # the K-1 interior knots
        d = (inputs[..., None] - knots).abs().min(dim=-1).values
        if include_bounds:     # the spline meets its linear tails C1 at +-B: the same two-valued second derivative
            d = torch.minimum(d, (inputs.abs() - tail_bound).abs())
        else:
            d = torch.where((inputs >= -tail_bound) & (inputs <= tail_bound), d, torch.full_like(d, float("inf")))
        captured.append((d / float(np.spacing(np.float32(tail_bound)))).detach())
        return real(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse=inverse,
                    tail_bound=tail_bound, min_bin_width=min_bin_width, **kw)

    mod.unconstrained_rational_quadratic_spline = spy
    try:
        oracle.double()
        with torch.no_grad():
            oracle.log_prob(theta.double(), x.double())
    finally:
        mod.unconstrained_rational_quadratic_spline = real
        oracle.float()
    return torch.cat([c.reshape(theta.shape[0], -1) for c in captured], dim=1)


# ---- one training pass: autograd through the oracle / the HIP kernels through the C ABI (shared by the GPU test modules)
def oracle_training_grad(oracle, est, theta, x, w=None, double=True):
    dt = torch.float64 if double else torch.float32
    oracle.double() if double else oracle.float()
    oracle.zero_grad()
    th = theta.to(dt).requires_grad_(True)
    xx = x.to(dt).requires_grad_(True)
    xe = xx if xx.shape[0] == th.shape[0] else xx.expand(th.shape[0], -1)
    l = oracle.loss(th, xe)
    ww = torch.full((th.shape[0],), 1.0 / th.shape[0], dtype=dt) if w is None else w.to(dt)
    (l * ww).sum().backward()
    named = dict(oracle.named_parameters())
    flat = torch.zeros(est.net.flat_params.numel(), dtype=dt)
    for key, off, n_, _ in est.net._slices():
        flat[off : off + n_] = named["net." + key].grad.reshape(-1)
    oracle.float()
    return l.detach(), flat, th.grad, xx.grad


def hip_training_pass(est, theta, x, w=None, want_gx=False):
    n = theta.shape[0]
    grad = torch.empty_like(est.net.flat_params.data)
    ws = train_workspace(est.net, n, "cuda")
    ws.fill_(float("nan"))
    gx = torch.full((n, x.shape[1]), float("nan"), device="cuda") if want_gx else None
    losses, gth = loss_fwd_bwd(est.net, theta.cuda().contiguous(), x.cuda().contiguous(),
                               None if w is None else w.cuda().contiguous(), 1.0 / n, grad, want_grad_theta=True,
                               workspace=ws, grad_x_out=gx)
    torch.cuda.synchronize()
    return losses.cpu(), grad.cpu(), gth.cpu(), (gx.cpu() if want_gx else None)
