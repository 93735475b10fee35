"""Properties that pin the oracle's restatement of nflows where no golden vector exists
(SURVEY.md section 8c): invertibility, log-det vs autograd Jacobian (fp64), identity
tails, near-identity initialisation, nflows-style state_dict layout (Appendix C)."""

import torch
import torch.autograd.functional as AF

from oracle.nsf_oracle import NSFOracle, LULinear, unconstrained_rational_quadratic_spline
from tests.helpers import linear_gaussian_data


def _oracle64(D=5, C=3, perturb=0.05, **kw):
    theta, x = linear_gaussian_data(500, D, C)
    torch.manual_seed(1)
    o = NSFOracle(theta, x, **kw).double()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for p in o.parameters():
            p.add_(perturb * torch.randn(p.shape, generator=g, dtype=torch.float64))
    return o, theta.double(), x.double()


def test_spline_inverse_and_logdet_fp64():
    torch.manual_seed(0)
    N, K = 512, 10
    x = torch.rand(N, 5, dtype=torch.float64) * 8 - 4
    uw, uh, ud = (torch.randn(N, 5, K, dtype=torch.float64), torch.randn(N, 5, K, dtype=torch.float64),
                  torch.randn(N, 5, K - 1, dtype=torch.float64))
    xr = x.clone().requires_grad_(True)
    y, ld = unconstrained_rational_quadratic_spline(xr, uw.clone(), uh.clone(), ud.clone(), tail_bound=3.0)
    (g,) = torch.autograd.grad(y.sum(), xr)
    assert (g.log() - ld).abs().max() < 1e-10
    xb, ldb = unconstrained_rational_quadratic_spline(y.detach(), uw.clone(), uh.clone(), ud.clone(), inverse=True,
                                                      tail_bound=3.0)
    assert (xb - x).abs().max() < 1e-9 and (ld.detach() + ldb).abs().max() < 1e-9
    outside = x.abs() > 3
    assert torch.equal(y.detach()[outside], x[outside]) and (ld.detach()[outside] == 0).all()
    # boundary derivative is 1: the map is C1 at +-tail_bound
    assert (g[(x.abs() - 3).abs() < 1e-3] - 1).abs().max() < 5e-2 if ((x.abs() - 3).abs() < 1e-3).any() else True


def test_flow_invertible_and_logdet_matches_jacobian():
    o, theta, x = _oracle64()
    noise = o.inverse_transform(theta[:64], x[:64])
    back, ld_inv = o.sample_from_noise(noise, x[:64])
    assert (back - theta[:64]).abs().max() < 1e-9
    e = o.net._embedding_net(x[:1])
    z = theta[:1]
    for t in o.net._transform._transforms:
        y, ld = t(z, e)
        J = AF.jacobian(lambda u: t(u[None], e)[0][0], z[0])
        assert abs(torch.linalg.slogdet(J)[1].item() - ld.item()) < 1e-9
        z = y.detach()


def test_lu_linear_roundtrip_and_identity_init():
    lu = LULinear(6).double()
    z = torch.randn(9, 6, dtype=torch.float64)
    y, ld = lu(z)
    assert (y - z).abs().max() < 1e-6 and ld.abs().max() < 1e-6    # identity_init=True
    with torch.no_grad():
        for p in lu.parameters():
            p.add_(0.3 * torch.randn_like(p))
    y, ld = lu(z)
    zb, ldb = lu.inverse(y)
    assert (zb - z).abs().max() < 1e-10 and (ld + ldb).abs().max() < 1e-12


def test_state_dict_layout_follows_nflows_names():
    theta, x = linear_gaussian_data(100, 10, 10)
    o = NSFOracle(theta, x)
    keys = list(o.state_dict().keys())
    assert keys[0] == "net._transform._transforms.0._shift" and keys[1].endswith("_scale")
    assert "net._transform._transforms.1.transform_net.initial_layer.weight" in keys
    assert "net._transform._transforms.2.unconstrained_upper_diag" in keys
    assert "net._embedding_net.0._mean" in keys and "net._embedding_net.0._std" in keys
    assert sum(p.numel() for p in o.parameters()) == 98025
    assert o.net._distribution._log_z.dtype == torch.float32        # flow.py:1486-1487
    sd = o.state_dict()
    assert sd["net._transform._transforms.1.transform_net.final_layer.weight"].shape == (145, 50)
    assert sd["net._transform._transforms.1.identity_features"].tolist() == [1, 3, 5, 7, 9]
    assert sd["net._transform._transforms.3.identity_features"].tolist() == [0, 2, 4, 6, 8]


def test_log_prob_shapes_and_self_consistency():
    """The reference's own numeric pins (tests/density_estimator_test.py:227-333)."""
    theta, x = linear_gaussian_data(100, 4, 7)
    torch.manual_seed(1)
    o = NSFOracle(theta, x)
    with torch.no_grad():
        lp = o.log_prob(theta[:10].unsqueeze(0).repeat(2, 1, 1), x[:10])
        assert lp.shape == (2, 10) and torch.allclose(lp[0], lp[1], rtol=1e-4)
        assert torch.allclose(o.log_prob(theta[:10], x[:10]), lp[:1], atol=1e-5)
        assert o.loss(theta[:10], x[:10]).shape == (10,)
        assert o.sample((3, 2), x[:5]).shape == (3, 2, 5, 4)
        lp1 = o.log_prob(theta[:10].unsqueeze(1), x[:1])
        assert lp1.shape == (10, 1)
