"""GPU parity of the FMPE vector-field kernels (csrc/fmpe.hip) through the C ABI:
 * against outputs of the real sbi classes (tests/golden/fmpe_reference.pt) and
 * against the CPU oracle (oracle/fmpe_oracle.py, itself pinned to those outputs) on more shapes:
per-row CFM loss, d(mean loss)/d(parameters) and the velocity ODE solvers integrate.
Tolerances: loss / velocity 2e-5 relative (fp32 MFMA vs fp32 CPU), gradients 3e-4 of the block maximum."""

import os

import pytest
import torch

from oracle.fmpe_oracle import FMPEOracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fmpe_reference.pt")


def make_pair(D, C, H=100, L=5, seed=0, n=512):
    """(oracle on CPU, sbi_amd estimator on GPU) with identical perturbed parameters, plus inputs."""
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    torch.manual_seed(seed)
    theta = torch.randn(n, D) * torch.linspace(0.5, 2.5, D) + torch.linspace(-1.0, 1.0, D)
    x = torch.randn(n, C) * 0.7 + theta[:, :1] * 0.5 + 0.3
    est = build_flow_matching_estimator(theta, x, hidden_features=H, num_layers=L)
    with torch.no_grad():
        est.net.flat_params.add_(0.05 * torch.randn_like(est.net.flat_params))
    oracle = FMPEOracle(D, C, H=H, L=L)
    oracle.load_reference_state_dict(est.net.reference_state_dict())
    times = torch.rand(n)
    noise = torch.randn(n, D)
    return oracle, est.cuda(), theta, x, times, noise


def flat_grad_of(oracle, est):
    out = torch.zeros(est.net.hyper.param_count())
    for key, off, cnt, _ in est.net.slices():
        out[off : off + cnt] = oracle.p[("net." + key).replace(".", "/")].grad.reshape(-1)
    return out


def check_grads(est, got, ref):
    scale = ref.abs().max().item()
    for key, off, cnt, _ in est.net.slices():
        a, b = got[off : off + cnt], ref[off : off + cnt]
        tol = 3e-4 * max(b.abs().max().item(), 1e-3 * scale) + 1e-8
        err = (a - b).abs().max().item()
        assert err <= tol, f"{key}: err {err:.3e} tol {tol:.3e} (block max {b.abs().max().item():.3e})"


@pytest.mark.parametrize("name", ["default_D5_C3", "H48_L2_D3_C4"])
def test_against_real_sbi_outputs(name):
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator, loss_fwd_bwd

    g = torch.load(GOLD, weights_only=False)[name]
    kw = g["kw"]
    est = build_flow_matching_estimator(g["theta"], g["x"], hidden_features=kw.get("hidden_features", 100),
                                        num_layers=kw.get("num_layers", 5))
    est.net.load_reference_state_dict(g["state"])
    est = est.cuda()
    n = g["theta"].shape[0]
    losses = est.loss(g["theta"].cuda(), g["x"].cuda(), times=g["times"].cuda(), noise=g["noise"].cuda())
    assert (losses.detach().cpu() - g["losses"]).abs().max() <= 2e-5 * g["losses"].abs().max()
    grad = torch.empty_like(est.net.flat_params.data)
    loss_fwd_bwd(est.net, g["theta"].cuda(), g["x"].cuda(), g["times"].cuda(), g["noise"].cuda(), None, 1.0 / n, grad)
    ref = torch.zeros(grad.numel())
    for key, off, cnt, _ in est.net.slices():
        ref[off : off + cnt] = g["grads"]["net." + key].reshape(-1)
    check_grads(est, grad.cpu(), ref)
    v = est(g["theta_q"].cuda(), g["x"][:1].cuda(), g["tq"].cuda())
    assert (v.cpu() - g["vel"]).abs().max() <= 2e-5 * g["vel"].abs().max()


SHAPES = [
    dict(D=50, C=50),                      # BASELINE configs[4] dimensions
    dict(D=5, C=3),
    dict(D=7, C=20, H=64, L=3),
    dict(D=3, C=4, H=128, L=2),
    dict(D=17, C=33, H=100, L=1),
    dict(D=1, C=1, H=32, L=2),
    dict(D=70, C=100, H=100, L=2),         # more than four 16-feature input blocks: the non-prefetched path
    dict(D=128, C=65, H=48, L=1),
]


@pytest.mark.parametrize("cfg", SHAPES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_loss_and_gradients_match_oracle(cfg):
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import loss_fwd_bwd, train_workspace

    oracle, est, theta, x, times, noise = make_pair(**cfg)
    n = 333   # ragged: not a multiple of 64
    th, xx, tt, nz = theta[:n], x[:n], times[:n], noise[:n]
    w = torch.linspace(0.5, 1.5, n) / n
    oracle.zero_grad()
    ref_losses = oracle.loss(th, xx, tt, nz)
    (ref_losses * w).sum().backward()
    ref = flat_grad_of(oracle, est)
    grad = torch.empty_like(est.net.flat_params.data)
    ws = train_workspace(est.net, n, "cuda")
    ws.fill_(float("nan"))     # nothing the kernels do not write themselves may reach the result
    losses = loss_fwd_bwd(est.net, th.cuda(), xx.cuda(), tt.cuda(), nz.cuda(), w.cuda(), 0.0, grad, workspace=ws)
    torch.cuda.synchronize()
    assert (losses.cpu() - ref_losses.detach()).abs().max() <= 2e-5 * ref_losses.abs().max()
    assert torch.isfinite(grad).all()
    check_grads(est, grad.cpu(), ref)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 64, 65, 200])
def test_velocity_matches_oracle_ragged_and_broadcast(n):
    oracle, est, theta, x, times, noise = make_pair(D=6, C=4, n=256)
    with torch.no_grad():
        ref = oracle.velocity(theta[:n], x[:n], times[:n])
        ref_b = oracle.velocity(theta[:n], x[:1], times[:1].expand(n))
    got = est(theta[:n].cuda(), x[:n].cuda(), times[:n].cuda()).cpu()
    got_b = est(theta[:n].cuda(), x[:1].cuda(), times[:1].cuda()).cpu()
    assert (got - ref).abs().max() <= 2e-5 * ref.abs().max()
    assert (got_b - ref_b).abs().max() <= 2e-5 * ref_b.abs().max()


def test_autograd_bridge_and_loss_without_grad():
    oracle, est, theta, x, times, noise = make_pair(D=4, C=3, H=64, L=2)
    n = 100
    args = (theta[:n].cuda(), x[:n].cuda())
    est.zero_grad()
    loss = est.loss(*args, times=times[:n].cuda(), noise=noise[:n].cuda())
    loss.mean().backward()
    oracle.zero_grad()
    oracle.loss(theta[:n], x[:n], times[:n], noise[:n]).mean().backward()
    check_grads(est, est.net.flat_params.grad.cpu(), flat_grad_of(oracle, est))
    with torch.no_grad():
        l2 = est.loss(*args, times=times[:n].cuda(), noise=noise[:n].cuda())
    assert torch.equal(l2, loss.detach())
    # draws made internally: finite, right shape
    l3 = est.loss(*args)
    assert l3.shape == (n,) and torch.isfinite(l3).all()


def test_full_batch_65536_is_deterministic_and_finite():
    """BASELINE configs[4] shape at the bench batch: two passes give bit-identical gradients."""
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import loss_fwd_bwd, train_workspace

    _, est, *_ = make_pair(D=50, C=50, n=64)
    n = 65536
    torch.manual_seed(3)
    th, xx = torch.randn(n, 50, device="cuda"), torch.randn(n, 50, device="cuda")
    tt, nz = torch.rand(n, device="cuda"), torch.randn(n, 50, device="cuda")
    ws = train_workspace(est.net, n, "cuda")
    g1, g2 = torch.empty_like(est.net.flat_params.data), torch.empty_like(est.net.flat_params.data)
    l1 = loss_fwd_bwd(est.net, th, xx, tt, nz, None, 1.0 / n, g1, workspace=ws)
    l2 = loss_fwd_bwd(est.net, th, xx, tt, nz, None, 1.0 / n, g2, workspace=ws)
    torch.cuda.synchronize()
    assert torch.isfinite(l1).all() and torch.isfinite(g1).all()
    assert torch.equal(l1, l2) and torch.equal(g1, g2)


@pytest.mark.parametrize("input_event", [1, 4])
@pytest.mark.parametrize("condition_event", [1, 7])
@pytest.mark.parametrize("batch_dim", [1, 10])
def test_shape_conventions_of_the_reference_suite(input_event, condition_event, batch_dim):
    """tests/vf_estimator_test.py:16-132 of the reference: `loss` -> (batch,), `forward` with batched and with
    scalar time -> (batch, *event)."""
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    torch.manual_seed(0)
    building_thetas = torch.randint(0, 4, (100, input_event), dtype=torch.float32)
    building_xs = torch.randn(100, condition_event)
    est = build_flow_matching_estimator(torch.randn_like(building_thetas), torch.randn_like(building_xs)).cuda()
    inputs, condition = building_thetas[:batch_dim].cuda(), building_xs[:batch_dim].cuda()
    losses = est.loss(inputs, condition=condition)
    assert losses.shape == (batch_dim,) and losses.is_cuda and torch.isfinite(losses).all()
    out = est(inputs, condition=condition, time=torch.rand(batch_dim).cuda())
    assert out.shape == (batch_dim, input_event)
    out = est(inputs, condition=condition, time=torch.rand(()).cuda())
    assert out.shape == (batch_dim, input_event) and torch.isfinite(out).all()


def test_structured_conditions_are_refused():
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    with pytest.raises(NotImplementedError):
        build_flow_matching_estimator(torch.randn(50, 4), torch.randn(50, 3, 3))
    with pytest.raises(NotImplementedError):
        build_flow_matching_estimator(torch.randn(50, 4), torch.randn(50, 3), embedding_net=torch.nn.Linear(3, 3))
