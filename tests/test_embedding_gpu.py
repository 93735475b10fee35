"""Trainable embedding net in front of the flow (flow.py:1395-1416): the kernels return d loss / d embedded x
(`grad_x_out`), autograd carries it into the embedding's weights."""

import warnings

import pytest
import torch
from torch import nn
from torch.distributions import MultivariateNormal

from sbi_amd.inference import NPE
from sbi_amd.neural_nets import NSFConfig
from sbi_amd.neural_nets.estimators.nsf_flow import loss_fwd_bwd
from sbi_amd.simulators.linear_gaussian import linear_gaussian, true_posterior_linear_gaussian_mvn_prior
from sbi_amd.utils.metrics import c2st
from tests.helpers import matched_pair
from tests.test_nsf_train_gpu import oracle_flat_grad

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [dict(D=10, C=10), dict(D=4, C=7), dict(D=3, C=20, hidden_features=32), dict(D=1, C=3),
                                 dict(D=5, C=3, num_transforms=2)],
                         ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_grad_wrt_condition_matches_autograd(cfg):
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    n = 333
    theta, x = theta_d[:n], x_d[:n]
    w = torch.linspace(0.5, 1.5, n)
    xo = x.clone().requires_grad_(True)
    oracle.zero_grad()
    (oracle.loss(theta, xo) * w).sum().backward()
    gref = oracle_flat_grad(oracle, est)
    grad = torch.empty_like(est.net.flat_params.data)
    gx = torch.full((n, x.shape[1]), float("nan"), device="cuda")
    loss_fwd_bwd(est.net, theta.cuda(), x.cuda(), w.cuda(), 0.0, grad, grad_x_out=gx)
    torch.cuda.synchronize()
    assert torch.isfinite(gx).all()
    scale = xo.grad.abs().max().item()
    err = (gx.cpu() - xo.grad).abs().max().item()
    print(f"d loss/d x: max|ref|={scale:.3e} err={err:.3e}")
    assert err <= 3e-4 * scale
    assert (grad.cpu() - gref).abs().max() <= 3e-4 * gref.abs().max()      # parameter gradient unchanged


def test_embedding_net_gradients_through_the_bridge():
    """est = Standardize -> MLP -> kernels; gradients of the MLP weights against the same composition on the oracle."""
    from sbi_amd.neural_nets.net_builders.flow import build_nsf
    from oracle.nsf_oracle import NSFOracle

    torch.manual_seed(0)
    n, D, Cx, Ce = 400, 4, 12, 6
    theta = torch.randn(n, D) * 0.5
    x = torch.randn(n, Cx) + theta.repeat(1, 3)
    emb = nn.Sequential(nn.Linear(Cx, 16), nn.Tanh(), nn.Linear(16, Ce))
    torch.manual_seed(1)
    est = build_nsf(theta, x, embedding_net=emb)
    assert est.net.hyper.C == Ce
    # oracle on the embedded features with the same flow weights
    with torch.no_grad():
        e_train = est.embedding_net(x)
    torch.manual_seed(1)
    oracle = NSFOracle(theta, e_train, z_score_theta="independent", z_score_x="none")
    oracle.load_state_dict({k: v for k, v in est.net.nflows_state_dict().items()}, strict=False)
    emb_cpu = est.embedding_net
    th, xx = theta[:200], x[:200]
    for p in emb_cpu.parameters():
        p.grad = None
    oracle.loss(th, emb_cpu(xx)).mean().backward()
    ref = [p.grad.clone() for p in emb_cpu.parameters() if p.requires_grad]
    for p in emb_cpu.parameters():
        p.grad = None
    est = est.to("cuda")
    est.loss(th.cuda(), xx.cuda()).mean().backward()
    got = [p.grad.cpu() for p in est.embedding_net.parameters() if p.requires_grad]
    assert len(got) == len(ref) == 4
    for a, b in zip(got, ref):
        assert (a - b).abs().max() <= 5e-4 * b.abs().max() + 1e-7
    # sample / log_prob shapes with a raw-x condition
    assert est.sample((5,), xx[:3].cuda()).shape == (5, 3, D)
    with torch.no_grad():
        assert est.log_prob(th[:7].cuda().unsqueeze(0), xx[:7].cuda()).shape == (1, 7)


def test_npe_with_trainable_embedding_net_c2st():
    """x is a redundant 12-D copy of a 2-D linear-Gaussian observation; a linear embedding has to find the 2-D
    summary (tests/embedding_net_test.py style)."""
    dim = 2
    torch.manual_seed(0)
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
    theta = prior.sample((3000,)).cpu()
    mix = torch.randn(dim, 12)
    x = linear_gaussian(theta, shift, cov) @ mix
    x_o = torch.zeros(1, dim) @ mix
    target = true_posterior_linear_gaussian_mvn_prior(torch.zeros(1, dim), shift, cov, torch.zeros(dim),
                                                      torch.eye(dim)).sample((1000,))
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator=NSFConfig(embedding_net=nn.Linear(12, 4)), device="cuda",
              show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=200)
    post = inf.build_posterior().set_default_x(x_o)
    samples = post.sample((1000,), show_progress_bars=False).cpu()
    score = c2st(samples, target).item()
    print(f"embedding NPE c2st={score:.3f} epochs={inf.summary['epochs_trained'][-1]}")
    assert 0.4 <= score <= 0.62


def _frozen_embedding_estimator(n=600, D=3, Cx=9, Ce=5):
    """Estimator with a FROZEN embedding net whose raw-x width differs from the embedded width."""
    from sbi_amd.neural_nets.net_builders.flow import build_nsf

    torch.manual_seed(0)
    theta = torch.randn(n, D) * 0.5
    x = torch.randn(n, Cx) + theta.repeat(1, 3)
    emb = nn.Sequential(nn.Linear(Cx, Ce), nn.Tanh())
    for p in emb.parameters():
        p.requires_grad_(False)
    torch.manual_seed(1)
    est = build_nsf(theta, x, embedding_net=emb)
    return est, theta, x


def test_fused_step_applies_a_frozen_embedding_net():
    """the fused trainer is chosen when the embedding has no trainable parameters; it must
    still APPLY standardize -> embedding before the kernels (it used to feed them raw x)."""
    from sbi_amd.inference.trainers.fused import FusedTrainStep

    est, theta, x = _frozen_embedding_estimator()
    est = est.cuda()
    th, xx = theta[:256].cuda(), x[:256].cuda()
    stepper = FusedTrainStep(est, distributed=False)
    fused_losses = stepper.loss_and_grad(th, xx)
    fused_grad = stepper.grad.clone()
    est.zero_grad()
    ref_losses = est.loss(th, xx)              # autograd bridge: embeds, then the same kernels
    ref_losses.mean().backward()
    assert (fused_losses - ref_losses.detach()).abs().max() <= 1e-6
    assert (fused_grad - est.net.flat_params.grad).abs().max() <= 1e-6 * est.net.flat_params.grad.abs().max() + 1e-9
    # NPE.train() picks the fused path for this estimator and its training / validation losses agree
    prior = MultivariateNormal(torch.zeros(3, device="cuda"), torch.eye(3, device="cuda"))
    emb = nn.Sequential(nn.Linear(9, 5), nn.Tanh())
    for p in emb.parameters():
        p.requires_grad_(False)
    inf = NPE(prior=prior, density_estimator=NSFConfig(embedding_net=emb), device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=100, max_num_epochs=3)
    assert inf._stepper is not None, "frozen embedding should train on the fused path"
    tr, va = inf.summary["training_loss"][-1], inf.summary["validation_loss"][-1]
    assert abs(tr - va) < 1.0, f"training ({tr}) and validation ({va}) losses disagree: x was not embedded"


def test_fused_path_with_structured_x_and_a_parameter_free_embedding():
    """x of shape (N, c, h, w) in front of `nn.Flatten` trains on the fused path; the device
    sampler only gathers flat rows, so this shape keeps the index path of the epoch loop (it used to raise
    'ShuffledGather expects theta (N, D) and x (N, C)')."""
    torch.manual_seed(0)
    n, D = 600, 3
    theta = torch.randn(n, D)
    x = (theta.repeat(1, 4) + 0.3 * torch.randn(n, 12)).reshape(n, 1, 3, 4)
    prior = MultivariateNormal(torch.zeros(D, device="cuda"), torch.eye(D, device="cuda"))
    inf = NPE(prior=prior, density_estimator=NSFConfig(embedding_net=nn.Flatten()), device="cuda",
              show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.append_simulations(theta, x).train(training_batch_size=100, max_num_epochs=4)
    assert inf._stepper is not None, "a parameter-free embedding should train on the fused path"
    assert est.condition_shape == torch.Size([1, 3, 4])
    tr, va = inf.summary["training_loss"], inf.summary["validation_loss"]
    assert len(tr) == 5 and tr[-1] < tr[0] and abs(tr[-1] - va[-1]) < 1.0
    lp = est.log_prob(theta[:7].cuda().unsqueeze(0), x[:7].cuda())
    assert lp.shape == (1, 7) and torch.isfinite(lp).all()


def test_mcmc_fused_potential_embeds_the_observation():
    """MCMCPosterior's fused potential must evaluate the flow on the EMBEDDED x_o."""
    from sbi_amd.inference.posteriors.mcmc_posterior import MCMCPosterior
    from sbi_amd.inference.potentials.posterior_based_potential import posterior_estimator_based_potential
    from sbi_amd.inference.posteriors.mcmc_posterior import unconstrained_potential
    from sbi_amd.utils.sbiutils import mcmc_transform

    est, theta, x = _frozen_embedding_estimator()
    est = est.cuda()
    prior = MultivariateNormal(torch.zeros(3, device="cuda"), torch.eye(3, device="cuda"))
    potential_fn, _ = posterior_estimator_based_potential(est, prior, x_o=None)
    tf = mcmc_transform(prior, device="cuda")
    post = MCMCPosterior(potential_fn, prior, tf, num_chains=8, thin=1, warmup_steps=2, device="cuda")
    x_o = x[:1].cuda()
    post.set_default_x(x_o)
    post.potential_fn.set_x(x_o, x_is_iid=True)
    fused = post._fused_potential()
    assert fused is not None
    u = torch.randn(64, 3, device="cuda")
    logp, lad = fused(u)
    ref = unconstrained_potential(post.potential_fn, tf, "cuda")(u)
    assert (logp - lad - ref).abs().max() <= 1e-4
    s = post.sample((40,), show_progress_bars=False)
    assert s.shape == (40, 3) and torch.isfinite(s).all()
