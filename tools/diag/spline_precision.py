"""Where the fp32 evaluation of an NSF log_prob loses its digits (CPU experiment on the oracle, no GPU needed).

Question: at theta-dim 10, 65 536 rows, 9 % of the rows of an fp32 `log_prob` -- the eager
oracle's and the kernels' alike -- are further than 1e-5 (absolute) from an fp64 evaluation.  Which part of the
arithmetic is responsible, and what would an implementation have to do to be better than eager fp32?

Method: the oracle's forward pass is re-run with ONE component at a time in fp32 and everything else in fp64 (and the
reverse), then with variants of the spline's normalisation.  Output (this container, seed 0, 16 384 rows):

  everything fp32                              9.3 % of rows beyond 1e-5, rms 6.0e-6
  fp32 terms, summed in fp64                   8.9 %                      5.9e-6   <- summation order is NOT it
  conditioner fp64, rest fp32                  8.7 %                      6.0e-6   <- nor the GEMMs
  spline fp64, rest fp32                       0.2 %                      2.3e-6   <- the spline is
  only conditioner fp32                        0.0 %                      2.4e-7
  only LULinear fp32                           0.2 %                      2.2e-6
  only the state rounded to fp32               0.0 %                      1.0e-6
  spline fp32, exp in fp32, but softmax sum / division / affine map / knot cumsum / bin extents in fp64
                                               0.4 %                      2.7e-6   <- what csrc/nsf_device.h precise_bin does
  same with the softmax sum left in fp32       3.9 %                      4.7e-6
  two-float knots only (widths by re-diff)     5.6 %                      5.2e-6

Reading: log|dy/dx| of a rational-quadratic bin contains log((h / w)^2 ...); every ~1e-7 relative error in the
selected bin's width or height enters doubled, and 25 spline evaluations per row accumulate.  nflows obtains w and h as
DIFFERENCES of rounded knot positions (cumsum -> affine -> overwrite ends -> re-diff), after an fp32 softmax.
"""
import copy
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import oracle.nsf_oracle as mod            # noqa: E402
from tests.helpers import matched_pair     # noqa: E402

N = 16384


def report(name, v, ref):
    d = (v - ref).abs()
    print(f"{name:58s} max {d.max().item():.2e}  beyond 1e-5: {(d > 1e-5).double().mean().item():7.2%}  "
          f"rms {d.pow(2).mean().sqrt().item():.2e}")


def mixed(oracle, o64, theta, x, cond_dt, spline_dt, lu_dt, state_dt, sum_dt=torch.float64):
    with torch.no_grad():
        e32, e64 = oracle.net._embedding_net(x), o64.net._embedding_net(x.double())
        out, total = theta.to(state_dt), torch.zeros(theta.shape[0], dtype=sum_dt)
        for t32, t64 in zip(oracle.net._transform._transforms, o64.net._transform._transforms):
            if isinstance(t32, mod.PiecewiseRationalQuadraticCouplingTransform):
                tc = t64 if cond_dt == torch.float64 else t32
                idn, tr = out[:, tc.identity_features].to(cond_dt), out[:, tc.transform_features].to(spline_dt)
                params = tc.transform_net(idn, e64 if cond_dt == torch.float64 else e32)
                if cond_dt == torch.float64 and spline_dt == torch.float32:
                    params = params.float()
                b, d = tr.shape
                ts, ld = tc._piecewise_cdf(tr, params.to(spline_dt).reshape(b, d, -1).clone(), False)
                new = out.clone()
                new[:, tc.transform_features] = ts.to(state_dt)
                out, total = new, total + ld.to(sum_dt).sum(1)
            elif isinstance(t32, mod.LULinear):
                o, ld = (t64 if lu_dt == torch.float64 else t32)(out.to(lu_dt))
                out, total = o.to(state_dt), total + ld.to(sum_dt)
            else:
                o, ld = (t64 if state_dt == torch.float64 else t32)(out)
                out, total = o, total + ld.to(sum_dt)
        return -0.5 * (out.double() ** 2).sum(1) - o64.net._distribution._log_z + total


VAR = {}


def rqs_variant(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse=False, left=0.,
                right=1., bottom=0., top=1., min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3):
    uw, uh, ud = unnormalized_widths, unnormalized_heights, unnormalized_derivatives
    K, f32 = uw.shape[-1], inputs.dtype

    def side(u, lo, hi, mn):
        e = torch.exp(u - u.max(-1, keepdim=True).values)                   # fp32 exp of an fp32 argument, always
        if VAR.get("norm") == "f64":
            p = e.double() / e.double().sum(-1, keepdim=True)
        elif VAR.get("norm") == "sum32":
            p = e.double() / e.sum(-1, keepdim=True).double()
        else:
            p = e / e.sum(-1, keepdim=True)
        w = mn + (1 - mn * K) * p
        cw = F.pad(torch.cumsum(w.double() if VAR.get("knots2") else w.to(f32), -1), (1, 0))
        cw = (hi - lo) * cw + lo
        cw[..., 0], cw[..., -1] = lo, hi
        c_hi = cw.to(f32)
        c_lo = (cw - c_hi.double()).to(f32) if VAR.get("knots2") else torch.zeros_like(c_hi)
        ext = (cw[..., 1:] - cw[..., :-1]).to(f32) if VAR.get("knots2") else c_hi[..., 1:] - c_hi[..., :-1]
        return c_hi, c_lo, ext

    cumw, cumw_lo, widths = side(uw, left, right, min_bin_width)
    cumh, cumh_lo, heights = side(uh, bottom, top, min_bin_height)
    derivatives = min_derivative + F.softplus(ud)
    bin_idx = mod.searchsorted(cumw, inputs)[..., None]
    g = lambda t: t.gather(-1, bin_idx)[..., 0]      # noqa: E731
    icw, icw_lo, ibw, ich, ich_lo = g(cumw), g(cumw_lo), g(widths), g(cumh), g(cumh_lo)
    delta = heights / widths
    idl, idv, idv1, ih = g(delta), g(derivatives), g(derivatives[..., 1:]), g(heights)
    th = ((inputs - icw) - icw_lo) / ibw
    tomt = th * (1 - th)
    num = ih * (idl * th.pow(2) + idv * tomt)
    den = idl + (idv + idv1 - 2 * idl) * tomt
    out = ich + (num / den + ich_lo)
    dn = idl.pow(2) * (idv1 * th.pow(2) + 2 * idl * tomt + idv * (1 - th).pow(2))
    return out, torch.log(dn) - 2 * torch.log(den)


def main():
    oracle, _, _, _ = matched_pair(D=10, C=10, device=None)
    o64 = copy.deepcopy(oracle).double()
    g = torch.Generator().manual_seed(0)
    theta = torch.randn(N, 10, generator=g) * (0.1 ** 0.5)
    x = theta + (0.1 ** 0.5) * torch.randn(N, 10, generator=g)
    f32, f64 = torch.float32, torch.float64
    ref = mixed(oracle, o64, theta, x, f64, f64, f64, f64)
    for name, args in [("everything fp32 (terms summed in fp64)", (f32, f32, f32, f32)),
                       ("everything fp32, summed in fp32", (f32, f32, f32, f32, f32)),
                       ("conditioner fp64, rest fp32", (f64, f32, f32, f32)), ("spline fp64, rest fp32", (f32, f64, f32, f32)),
                       ("LULinear fp64, rest fp32", (f32, f32, f64, f32)), ("only the conditioner in fp32", (f32, f64, f64, f64)),
                       ("only the spline in fp32", (f64, f32, f64, f64)), ("only LULinear in fp32", (f64, f64, f32, f64)),
                       ("only the state rounded to fp32", (f64, f64, f64, f32))]:
        report(name, mixed(oracle, o64, theta, x, *args), ref)
    real = mod.rational_quadratic_spline
    for name, var in [("spline variant = nflows (control)", {}),
                      ("two-float knots, extents from them; fp32 softmax", {"knots2": True}),
                      ("softmax sum fp32, division / affine / knots fp64", {"knots2": True, "norm": "sum32"}),
                      ("softmax sum, division, affine, knots, extents in fp64", {"knots2": True, "norm": "f64"})]:
        VAR.clear()
        VAR.update(var)
        mod.rational_quadratic_spline = rqs_variant
        try:
            with torch.no_grad():
                v = oracle.log_prob(theta, x)[0].double()
        finally:
            mod.rational_quadratic_spline = real
        report(name, v, ref)


if __name__ == "__main__":
    main()
