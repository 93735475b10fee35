"""VectorFieldPosterior.log_prob on the GPU (sbi/inference/posteriors/vector_field_posterior.py:467-504): the velocity +
exact-Jacobian-trace kernel (csrc/fmpe.hip::fm_div_kernel behind sbi_amd_fmpe_velocity_div) against the real estimator's
autograd trace (tests/golden/fmpe_reference.pt, key `div`) and the CPU oracle, and the log-density of the
probability-flow ODE against the oracle's fp64 fixed-grid solve.

Tolerances: trace 3e-5 of max(|trace|, 1) against the fp64 oracle (fp32 MFMA chain of ~10 layers); log_prob 2e-3
absolute (adaptive fp32 Dormand-Prince at sbi's atol 1e-6 / rtol 1e-5 against RK4 with 48 fp64 steps)."""

import os

import pytest
import torch

from oracle.fmpe_oracle import FMPEOracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fmpe_reference.pt")


def make_pair(D, C, H=100, L=5, seed=0, n=256, scale=0.05):
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    torch.manual_seed(seed)
    theta = torch.randn(n, D) * torch.linspace(0.5, 2.5, D) + torch.linspace(-1.0, 1.0, D)
    x = torch.randn(n, C) * 0.7 + theta[:, :1] * 0.5 + 0.3
    est = build_flow_matching_estimator(theta, x, hidden_features=H, num_layers=L)
    with torch.no_grad():
        est.net.flat_params.add_(scale * torch.randn_like(est.net.flat_params))
    o64 = FMPEOracle(D, C, H=H, L=L).double()
    o64.load_reference_state_dict({k: v.double() for k, v in est.net.reference_state_dict().items()})
    return o64, est.cuda(), theta, x


@pytest.mark.parametrize("name", ["default_D5_C3", "H48_L2_D3_C4"])
def test_trace_against_real_sbi_autograd(name):
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    g = torch.load(GOLD, weights_only=False)[name]
    kw = g["kw"]
    est = build_flow_matching_estimator(g["theta"], g["x"], hidden_features=kw.get("hidden_features", 100),
                                        num_layers=kw.get("num_layers", 5))
    est.net.load_reference_state_dict(g["state"])
    est = est.cuda()
    v, div = est.ode_fn_and_divergence(g["theta_q"].cuda(), g["x"][:1].cuda(), g["tq"].cuda())
    assert (v.cpu() - g["vel"]).abs().max() <= 2e-5 * g["vel"].abs().max()
    assert (div.cpu() - g["div"]).abs().max() <= 3e-5 * max(g["div"].abs().max().item(), 1.0)


SHAPES = [
    dict(D=5, C=3),                        # sbi's default net
    dict(D=10, C=10),
    dict(D=1, C=2, H=32, L=1),
    dict(D=15, C=7, H=64, L=3),            # all fifteen tangent columns in use
    dict(D=16, C=5, H=100, L=2),           # two passes: 15 + 1 directions
    dict(D=31, C=33, H=48, L=1),           # three passes, second input / output block
    dict(D=50, C=50),                      # BASELINE configs[4] dimensions: four passes
    dict(D=40, C=20, H=128, L=2),          # eight hidden blocks
]


@pytest.mark.parametrize("cfg", SHAPES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_velocity_and_trace_match_oracle(cfg):
    o64, est, theta, x = make_pair(**cfg)
    D = cfg["D"]
    for n, per_row_x, per_row_t in [(1, False, False), (7, True, True), (8, False, True), (9, True, False),
                                    (100, True, True), (256, False, False)]:
        th = theta[:n] * 0.8 + 0.1
        xx = x[:n] if per_row_x else x[:1]
        tt = torch.linspace(0.0, 1.0, n) if per_row_t else torch.tensor([0.37])
        v, div = est.ode_fn_and_divergence(th.cuda(), xx.cuda(), tt.cuda())
        rv, rdiv = o64.velocity_and_divergence(th.double(), xx.double(), tt.double().expand(n))
        assert v.shape == (n, D) and div.shape == (n,)
        ev = (v.cpu().double() - rv).abs().max().item()
        ed = (div.cpu().double() - rdiv).abs().max().item()
        assert ev <= 3e-5 * max(rv.abs().max().item(), 1.0), (n, ev)
        assert ed <= 3e-5 * max(rdiv.abs().max().item(), 1.0), (n, ed, rdiv.abs().max().item())
        # the primal column computes exactly what the velocity kernel computes
        v_plain = est(th.cuda(), xx.cuda(), tt.cuda())
        assert (v - v_plain).abs().max().item() <= 1e-6 * max(rv.abs().max().item(), 1.0)
    # the trace is not small because the field is flat: it moves by O(1) across the batch
    assert rdiv.abs().max().item() > 1e-3


def _posterior(est, D, low=-30.0, high=30.0):
    from sbi_amd.inference.posteriors.vector_field_posterior import VectorFieldPosterior
    from sbi_amd.utils.torchutils import BoxUniform

    prior = BoxUniform(torch.full((D,), low).cuda(), torch.full((D,), high).cuda())
    return VectorFieldPosterior(est, prior)


@pytest.mark.parametrize("cfg", [dict(D=5, C=3), dict(D=2, C=4, H=48, L=2), dict(D=16, C=6, H=48, L=1)],
                         ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_log_prob_matches_fp64_solve_of_the_oracle(cfg):
    o64, est, theta, x = make_pair(scale=0.1, **cfg)
    D = cfg["D"]
    post = _posterior(est, D).set_default_x(x[:1])
    th = theta[:12]
    lp = post.log_prob(th)
    ref = o64.log_prob(th.double(), x[:1].double(), steps=48)      # RK4: (1/48)^4 ~ 2e-7
    assert lp.shape == (12,) and torch.isfinite(lp).all()
    err = (lp.cpu().double() - ref).abs().max().item()
    print(f"{cfg}: log_prob in [{ref.min().item():.3f}, {ref.max().item():.3f}], max |device - fp64 oracle| {err:.2e}")
    assert err <= 2e-3
    # tighter solver tolerances get closer; an explicit x equals the default one; 1-D theta is one row
    lp_tight = post.log_prob(th, ode_kwargs=dict(atol=1e-8, rtol=1e-7))
    assert (lp_tight.cpu().double() - ref).abs().max().item() <= max(0.5 * err, 2e-4)
    assert torch.equal(post.log_prob(th[:5], x=x[:1]), post.log_prob(th[:5]))
    assert post.log_prob(th[0]).shape == (1,)
    # chunked evaluation is the same computation per row (the controller sees other rows: solver-tolerance close)
    lp_chunked = post.log_prob(th, max_batch_size=5)
    assert (lp_chunked - lp).abs().max().item() <= 1e-3


def test_log_prob_is_minus_inf_outside_the_prior_and_validates_arguments():
    o64, est, theta, x = make_pair(D=3, C=2, H=32, L=1)
    post = _posterior(est, 3, low=-1.0, high=1.0).set_default_x(x[:1])
    th = torch.tensor([[0.1, 0.2, -0.3], [0.5, 2.0, 0.0], [-0.9, 0.9, 0.99]])
    lp = post.log_prob(th)
    assert torch.isfinite(lp[0]) and torch.isfinite(lp[2]) and lp[1] == float("-inf")
    with pytest.raises(NotImplementedError):
        post.log_prob(th, track_gradients=True)
    with pytest.raises(NotImplementedError):
        post.log_prob(th, ode_kwargs=dict(exact=False))
    with pytest.raises(TypeError):
        post.log_prob(th, ode_kwargs=dict(method="rk4"))
    with pytest.raises(ValueError):
        post.log_prob(torch.zeros(4, 2))
    assert post.log_prob(torch.zeros(0, 3)).shape == (0,)


def test_iid_observations_sum_the_flows_and_subtract_the_prior():
    """x with several rows = iid observations (vector_field_potential.py:175-192): sum_i log p(theta | x_i) - (n - 1) log
    p(theta); one row of x behaves as before; no prior is an error."""
    from sbi_amd.inference.posteriors.vector_field_posterior import VectorFieldPosterior

    o64, est, theta, x = make_pair(D=3, C=2, H=32, L=1)
    post = _posterior(est, 3)
    th = theta[:20] * 0.5
    xs = x[:3]
    singles = torch.stack([post.log_prob(th, x=xs[i : i + 1]) for i in range(3)])
    iid = post.log_prob(th, x=xs)
    want = singles.sum(0) - 2.0 * post.prior.log_prob(th.cuda())
    assert (iid - want).abs().max().item() <= 1e-4
    assert torch.equal(post.log_prob(th, x=xs[:1]), singles[0])
    with pytest.raises(AssertionError):
        VectorFieldPosterior(est, None).log_prob(th, x=xs)


def test_density_integrates_to_one_in_one_dimension():
    """theta-dim 1: exp(log_prob) over a wide grid integrates to one -- sign of the log-det and direction of the solve."""
    o64, est, theta, x = make_pair(D=1, C=2, H=32, L=2, scale=0.3)
    post = _posterior(est, 1).set_default_x(x[:1])
    grid = torch.linspace(-14.0, 14.0, 4001)
    lp = post.log_prob(grid[:, None])
    mass = torch.trapezoid(lp.double().exp().cpu(), grid.double()).item()
    assert abs(mass - 1.0) <= 3e-3, mass


def test_full_size_round_trip_sample_then_log_prob():
    """65 536 rows (BASELINE's batch): draw theta = flow(eps) with the sampling ODE (t: 1 -> 0), then log_prob's ODE
    (t: 0 -> 1) must land on eps again, and log_prob(theta) = log N(eps; 0, I) + ladj with a ladj that the oracle
    reproduces on a few of those rows."""
    o64, est, theta, x = make_pair(D=5, C=3, scale=0.1)
    post = _posterior(est, 5).set_default_x(x[:1])
    n = 65536
    g = torch.Generator(device="cuda").manual_seed(1)
    eps = torch.randn(n, 5, device="cuda", generator=g)
    from sbi_amd.samplers.ode_solvers import odeint_dopri5

    xo = x[:1].cuda()
    th = odeint_dopri5(lambda t, y: est.ode_fn(y, xo, t), eps.contiguous(), est.t_max, est.t_min, atol=1e-6, rtol=1e-5)
    lp = post.log_prob(th)
    assert lp.shape == (n,) and torch.isfinite(lp).all()
    # end point of the forward solve
    y0 = torch.cat([th.reshape(-1), torch.zeros(n, device="cuda")])

    def rhs(t, y):
        out = torch.empty_like(y)
        est.ode_fn_and_divergence(y[: n * 5].view(n, 5), xo, t, v_out=out[: n * 5].view(n, 5), div_out=out[n * 5 :])
        return out

    y1 = odeint_dopri5(rhs, y0, est.t_min, est.t_max, atol=1e-6, rtol=1e-5)
    back = y1[: n * 5].view(n, 5)
    assert (back - eps).abs().max().item() <= 2e-3
    base = (-0.5 * back ** 2).sum(-1) - 2.5 * 1.8378770664093453
    assert (lp - (base + y1[n * 5 :])).abs().max().item() <= 1e-4
    idx = torch.tensor([0, 777, 65535])
    ref = o64.log_prob(th[idx.cuda()].cpu().double(), x[:1].double(), steps=48)
    assert (lp[idx.cuda()].cpu().double() - ref).abs().max().item() <= 2e-3
