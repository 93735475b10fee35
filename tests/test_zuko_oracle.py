"""CPU checks of the zuko_nsf path: the oracle (oracle/zuko_oracle.py) by properties in fp64 (autoregressive in
both orders, log-det == Jacobian log-det, invertibility, identity outside [-5, 5], MaskedMLP masks known answers) and
the host mirror (layout == C ABI, masks == the oracle's, identical initialisation, builder / config / factory)."""
import pytest
import torch

from oracle.zuko_oracle import ZukoNSFOracle, masked_mlp_masks, rqs_forward, rqs_inverse, rqs_params
from sbi_amd import _lib
from sbi_amd.neural_nets import ZukoNSFConfig, posterior_nn
from sbi_amd.neural_nets.estimators.zuko_flow import ZukoHyper, ZukoNSFFlow
from sbi_amd.neural_nets.net_builders.flow import build_zuko_nsf


def _data(n=300, D=3, C=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    theta = torch.randn(n, D, generator=g) * 0.7 + 0.2
    return theta, theta[:, :1] * 0.5 + torch.randn(n, C, generator=g)


def test_masked_mlp_masks_known_answers():
    # 3 features x 2 outputs each, 2 context columns, order arange: feature i sees features < i and the context
    adj = torch.arange(3).repeat_interleave(2)[:, None] > torch.cat((torch.arange(3), torch.full((2,), -1)))
    m = masked_mlp_masks(adj, [5, 5])
    assert [r.int().tolist() for r in m[0]] == [[0, 0, 0, 1, 1], [1, 0, 0, 1, 1], [1, 1, 0, 1, 1], [0, 0, 0, 1, 1],
                                                [1, 0, 0, 1, 1]]                # unit u takes pattern u % 3
    assert m[1][0].int().tolist() == [1, 0, 0, 1, 0] and m[1][2].int().tolist() == [1, 1, 1, 1, 1]
    assert m[2][0].int().tolist() == m[2][1].int().tolist() == [1, 0, 0, 1, 0]  # both outputs of feature 0
    assert m[2][4].all()
    # composite dependency: output of feature i never depends on an input >= i
    dep = (m[2].double() @ m[1].double() @ m[0].double()) > 0
    for i in range(3):
        assert not dep[2 * i, i:3].any() and dep[2 * i, 3:].all()


def test_spline_is_monotone_invertible_and_identity_outside_the_box():
    torch.manual_seed(0)
    w, h, d = torch.randn(64, 8).double() * 2, torch.randn(64, 8).double() * 2, torch.randn(64, 7).double()
    hor, ver, der = rqs_params(w, h, d)
    assert torch.allclose(hor[:, 0], torch.full((64,), -5.0).double()) and torch.allclose(hor[:, -1],
                                                                                           torch.full((64,), 5.0).double())
    x = torch.linspace(-7, 7, 64).double()
    y, ladj = rqs_forward(x, hor, ver, der)
    outside = (x <= -5) | (x > 5)
    assert torch.equal(y[outside], x[outside]) and (ladj[outside] == 0).all()
    assert (rqs_inverse(y, hor, ver, der) - x).abs().max() < 1e-9
    xg = x.clone().requires_grad_(True)
    yg, ladj_g = rqs_forward(xg, hor, ver, der)
    (g,) = torch.autograd.grad(yg.sum(), xg)
    assert torch.allclose(torch.log(g), ladj_g, atol=1e-9) and (g > 0).all()


def test_oracle_flow_properties():
    theta, x = _data()
    torch.manual_seed(1)
    o = ZukoNSFOracle(theta, x, num_transforms=3, hidden_features=16).double()
    with torch.no_grad():
        for p in o.parameters():
            p.add_(0.3 * torch.randn_like(p))
    z, c = torch.randn(1, 3, dtype=torch.float64), torch.randn(1, 2, dtype=torch.float64)
    for idx, tri in ((1, "triu"), (2, "tril")):        # order arange, then reversed
        t = o.transforms[idx]
        J = torch.autograd.functional.jacobian(lambda a: t(a, c)[0], z)[0, :, 0, :]
        off = J.triu(1) if tri == "triu" else J.tril(-1)
        assert off.abs().max() == 0
        assert torch.allclose(torch.log(J.diagonal().abs()).sum(), t(z, c)[1][0], atol=1e-10)
    th, xx = theta[:64].double(), x[:64].double()
    noise = o.inverse_transform(th, xx)
    back, ld = o.sample_from_noise(noise, xx)
    assert (back - th).abs().max() < 1e-9
    import math

    base = (-0.5 * noise**2 - 0.5 * math.log(2 * math.pi)).sum(-1)
    assert torch.allclose(o.log_prob(th, xx)[0], base - ld, atol=1e-9)
    with torch.no_grad():
        assert o.float().sample((7,), x[:4]).shape == (7, 4, 3)


def test_host_mirror_layout_masks_and_init():
    lib = _lib.load()
    for kw in (dict(D=3, C=2, hidden_features=16, num_transforms=3, num_hidden_layers=3), dict(D=10, C=10),
               dict(D=1, C=4, num_hidden_layers=2), dict(D=16, C=32, hidden_features=64, num_bins=8)):
        h = ZukoHyper(**kw)
        assert lib.sbi_amd_maf_param_count(h.c_config()) == h.param_count(), kw
    assert lib.sbi_amd_maf_param_count(ZukoHyper(D=3, C=2, num_hidden_layers=6).c_config()) == _lib.E_UNSUPPORTED
    theta, x = _data()
    torch.manual_seed(4)
    est = build_zuko_nsf(theta, x, hidden_features=16, num_transforms=3)
    torch.manual_seed(4)
    o = ZukoNSFOracle(theta, x, hidden_features=16, num_transforms=3)
    assert isinstance(est, ZukoNSFFlow) and est.net.hyper.num_hidden_layers == 3   # sbi: [H] * num_transforms
    mine, ref = est.net.zuko_state_dict(), o.state_dict()
    assert set(mine) <= set(ref) and all(torch.equal(mine[k], ref[k]) for k in mine)
    est.net.load_zuko_state_dict(ref)                  # raises if the derived adjacency masks differ from the oracle's
    bad = dict(ref)
    bad["transforms.1.hyper.0.mask"] = ~ref["transforms.1.hyper.0.mask"]
    with pytest.raises(ValueError, match="adjacency mask"):
        est.net.load_zuko_state_dict(bad)


def test_config_factory_and_refusals():
    theta, x = _data()
    assert ZukoNSFConfig(num_bins=8).build(theta, x).net.hyper.num_bins == 8
    assert isinstance(posterior_nn("zuko_nsf", hidden_features=20, num_transforms=2)(theta, x), ZukoNSFFlow)
    with pytest.raises(NotImplementedError):
        build_zuko_nsf(theta, x, passes=2)
    with pytest.raises(NotImplementedError):
        build_zuko_nsf(theta, x, z_score_x="transform_to_unconstrained")
    with pytest.raises(NotImplementedError):
        build_zuko_nsf(theta, x, num_transforms=7)     # seven hidden layers per hyper-net
    build_zuko_nsf(theta, x, num_blocks=3, tail_bound=3.0)   # nflows-only kwargs are dropped, as build_zuko_flow does
