"""CPU checks of the maf_rqs path that need no GPU: the oracle (oracle/maf_oracle.py) by properties in fp64
(autoregressive structure, log-det == Jacobian log-det, invertibility, known-answer degree masks), the host-side
mirror (flat layout == C ABI offsets, nflows state-dict exchange, identical initialisation to the oracle's nflows
construction order, builder / config / factory API)."""
import pytest
import torch

from oracle.maf_oracle import MADE, MAFRQSOracle, MaskedLinear, _get_input_degrees
from sbi_amd import _lib
from sbi_amd.neural_nets import MAFRQSConfig, posterior_nn
from sbi_amd.neural_nets.estimators.maf_flow import MAFHyper, MAFRQSFlow
from sbi_amd.neural_nets.net_builders.flow import build_maf_rqs


def _data(n=300, D=4, C=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    theta = torch.randn(n, D, generator=g) * 0.8 + 0.3
    x = theta[:, :1] * 0.5 + torch.randn(n, C, generator=g)
    return theta, x


def test_made_degree_masks_known_answers():
    # D = 3, H = 5: hidden degrees i % 2 + 1 = [1, 2, 1, 2, 1]; inputs 1, 2, 3; outputs repeat(1..3, P)
    lin = MaskedLinear(_get_input_degrees(3), 5, 3, is_output=False)
    assert lin.degrees.tolist() == [1, 2, 1, 2, 1]
    assert lin.mask.tolist() == [[1, 0, 0], [1, 1, 0], [1, 0, 0], [1, 1, 0], [1, 0, 0]]
    out = MaskedLinear(lin.degrees, 6, 3, is_output=True)
    assert out.degrees.tolist() == [1, 1, 2, 2, 3, 3]
    assert out.mask[0].tolist() == [0, 0, 0, 0, 0]            # dim 0 sees nothing but the context
    assert out.mask[2].tolist() == [1, 0, 1, 0, 1] and out.mask[4].tolist() == [1, 1, 1, 1, 1]
    h = MAFHyper(D=3, C=2, hidden_features=5, num_bins=4)
    assert torch.equal(h.mask(0), lin.mask)
    assert torch.equal(h.mask(3)[:: 3 * 4 - 1][:3], MaskedLinear(lin.degrees, 3 * 11, 3, True).mask[::11][:3])
    # D = 1: hidden degree 0 everywhere, the single output (degree 1) sees every hidden unit, no input is used
    assert MAFHyper(D=1, C=2, hidden_features=4).mask(0).sum() == 0
    assert MAFHyper(D=1, C=2, hidden_features=4).mask(3).all()


def test_oracle_is_autoregressive_invertible_and_its_logdet_is_the_jacobians():
    theta, x = _data()
    torch.manual_seed(1)
    o = MAFRQSOracle(theta, x, num_transforms=3).double()
    with torch.no_grad():
        for p in o.parameters():
            p.add_(0.3 * torch.randn_like(p))
    t = o.net._transform._transforms[1]
    z, c = torch.randn(1, 4, dtype=torch.float64), torch.randn(1, 3, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(lambda a: t(a, c)[0], z)[0, :, 0, :]
    assert J.triu(1).abs().max() == 0                       # output i depends on inputs <= i only
    assert torch.allclose(torch.log(J.diagonal().abs()).sum(), t(z, c)[1][0], atol=1e-10)
    th, xx = theta[:64].double(), x[:64].double()
    noise = o.inverse_transform(th, xx)
    back, ld = o.sample_from_noise(noise, xx)
    assert (back - th).abs().max() < 1e-6   # (the bin search bumps the last knot by 1e-6)
    # whole flow: log_prob == base(noise) + logabsdet, and logabsdet(inverse) = -logabsdet(forward)
    lp = o.log_prob(th, xx)[0]
    base = -0.5 * (noise**2).sum(1) - o.net._distribution._log_z.double()
    assert torch.allclose(lp, base - ld, atol=1e-6)


def test_reference_self_consistency_checks():
    """The reference's own estimator tests (tests/density_estimator_test.py:227-333) on the oracle."""
    theta, x = _data(D=3, C=5)
    o = MAFRQSOracle(theta, x)
    with torch.no_grad():
        s = o.sample((7,), x[:4])
        assert s.shape == (7, 4, 3)
        lp_b = o.log_prob(s, x[:4])
        lp_l = torch.stack([o.log_prob(s[:, i : i + 1], x[i : i + 1])[:, 0] for i in range(4)], dim=1)
        assert torch.allclose(lp_b, lp_l, atol=1e-5)


def test_flat_layout_matches_the_c_abi_and_nflows_keys():
    lib = _lib.load()
    for kw in (dict(D=4, C=3), dict(D=10, C=10), dict(D=1, C=2, hidden_features=20, num_blocks=1),
               dict(D=7, C=5, hidden_features=64, num_bins=8, num_transforms=3, num_blocks=3)):
        h = MAFHyper(**kw)
        cfg = h.c_config()
        assert lib.sbi_amd_maf_param_count(cfg) == h.param_count()
        assert lib.sbi_amd_maf_packed_floats(cfg) > h.param_count()
        off = 0
        for t in range(h.num_transforms):
            for i, (key, shape, _) in enumerate(h.layer_entries()):
                assert lib.sbi_amd_maf_param_offset(cfg, t, i // 2, i % 2) == off, (kw, t, key)
                off += int(torch.Size(shape).numel())
    assert lib.sbi_amd_maf_param_count(MAFHyper(D=17, C=3).c_config()) == _lib.E_UNSUPPORTED
    assert lib.sbi_amd_maf_param_count(MAFHyper(D=4, C=3, num_bins=7).c_config()) == _lib.E_UNSUPPORTED
    assert lib.sbi_amd_maf_log_prob(MAFHyper(D=4, C=3).c_config(), None, None, None, None, 4, 4, None, None,
                                    None) == _lib.E_BADARG


def test_builder_matches_the_oracles_construction_and_exchanges_weights():
    theta, x = _data()
    torch.manual_seed(5)
    est = build_maf_rqs(theta, x)
    torch.manual_seed(5)
    o = MAFRQSOracle(theta, x)
    assert isinstance(est, MAFRQSFlow) and est.net.hyper.param_count() == sum(p.numel() for p in o.parameters())
    mine, ref = est.net.nflows_state_dict(), o.state_dict()
    assert set(mine) == set(ref)                              # exactly nflows' keys, mask / degrees buffers included
    o.load_state_dict(mine, strict=True)                      # ... so a strict load into the nflows-shaped module works
    for k in mine:                                            # ... and the same seed gives the same init
        assert torch.equal(mine[k].to(ref[k].dtype), ref[k]), k
    with torch.no_grad():
        for p in o.parameters():
            p.add_(0.1 * torch.randn_like(p))
    est.net.load_nflows_state_dict(o.state_dict())
    back = est.net.nflows_state_dict()
    assert all(torch.equal(back[k].to(ref[k].dtype), o.state_dict()[k]) for k in back)
    bad = dict(o.state_dict())                                # a checkpoint with other masks must not load silently
    key = next(k for k in bad if k.endswith("initial_layer.mask"))
    bad[key] = 1.0 - bad[key]
    with pytest.raises(ValueError, match="degree masks"):
        est.net.load_nflows_state_dict(bad)


def test_config_and_factory_surface():
    theta, x = _data()
    est = MAFRQSConfig(hidden_features=32, num_transforms=2, num_bins=8).build(theta, x)
    assert isinstance(est, MAFRQSFlow) and est.net.hyper.hidden_features == 32 and est.net.hyper.num_bins == 8
    assert repr(MAFRQSConfig(num_bins=5)) == "MAFRQSConfig(num_bins=5)"
    est2 = posterior_nn("maf_rqs", hidden_features=20, num_blocks=1)(theta, x)
    assert est2.net.hyper.num_blocks == 1 and est2.input_shape == torch.Size([4])
    with pytest.raises(NotImplementedError):
        build_maf_rqs(theta, x, tails=None)
    with pytest.raises(NotImplementedError):
        build_maf_rqs(theta, x, use_batch_norm=True)
    with pytest.raises(ValueError):                            # flow.py:275-280
        build_maf_rqs(theta, x, z_score_x="transform_to_unconstrained")
    with pytest.raises(RuntimeError, match="ROCm device"):     # no CPU fallback
        est.log_prob(theta[:3].unsqueeze(0), x[:3])


def test_both_readings_of_the_made_sqrt_hidden_question_are_evaluated():
    """The one recalled nflows detail this path cannot check offline: nflows'
    MaskedPiecewiseRationalQuadraticAutoregressiveTransform divides the width / height logits by sqrt(hidden_features)
    `if hasattr(self.autoregressive_net, "hidden_features")`; the oracle assumes MADE defines no such attribute
    (`scale_by_sqrt_hidden=False`).  Both readings exist in the oracle AND in the kernels' config; this pins what
    each reading MEANS by a known answer, so that the day a real nflows is importable
    (`tools/compare_with_nflows.py`, which evaluates both) the answer is a one-word change of a default:
    reading True == reading False with the width / height rows of every final layer divided by sqrt(H)."""
    import math

    theta, x = _data()
    H, K, D = 12, 5, 4
    kw = dict(hidden_features=H, num_transforms=2, num_bins=K)
    torch.manual_seed(3)
    plain = MAFRQSOracle(theta, x, scale_by_sqrt_hidden=False, **kw).double()
    with torch.no_grad():
        for p in plain.parameters():
            p.add_(0.3 * torch.randn_like(p))
    scaled = MAFRQSOracle(theta, x, scale_by_sqrt_hidden=True, **kw).double()
    scaled.load_state_dict(plain.state_dict())
    th, xx = theta[:50].double(), x[:50].double()
    lp_plain, lp_scaled = plain.log_prob(th, xx)[0], scaled.log_prob(th, xx)[0]
    assert (lp_plain - lp_scaled).abs().max() > 1e-2          # the switch is live
    # known answer: fold 1 / sqrt(H) into the final layers' width / height rows of the unscaled flow
    folded = MAFRQSOracle(theta, x, scale_by_sqrt_hidden=False, **kw).double()
    folded.load_state_dict(plain.state_dict())
    P = 3 * K - 1
    rows = torch.tensor([d * P + k for d in range(D) for k in range(2 * K)])
    with torch.no_grad():
        for name, p in folded.named_parameters():
            if "final_layer" in name:
                p[rows] /= math.sqrt(H)
    assert torch.allclose(folded.log_prob(th, xx)[0], lp_scaled, atol=1e-10)
    # the scaled reading is a proper flow too
    noise = scaled.inverse_transform(th, xx)
    back, _ = scaled.sample_from_noise(noise, xx)
    assert (back - th).abs().max() < 1e-6
    # and the product's config carries the same switch to the kernels (field 11 of sbi_amd_maf_config)
    assert MAFHyper(D=4, C=3, scale_by_sqrt_hidden=True).c_config().scale_by_sqrt_hidden == 1
    assert MAFHyper(D=4, C=3).c_config().scale_by_sqrt_hidden == 0
