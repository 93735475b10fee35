"""Multi-round NPE-C on the GPU path: the fused atomic step (train_forward -> softmax weights ->
train_backward on the same stash) against autograd through the CPU oracle with identical atoms, and a
two-round run on the linear-Gaussian task scored by C2ST (tests/linearGaussian_snpe_test.py:376-497)."""

import warnings

import pytest
import torch
from torch.distributions import MultivariateNormal

from sbi_amd.inference import NPE
from sbi_amd.inference.trainers.fused import FusedTrainStep
from sbi_amd.inference.trainers.npe import atomic
from sbi_amd.neural_nets import NSFConfig
from sbi_amd.simulators.linear_gaussian import linear_gaussian, true_posterior_linear_gaussian_mvn_prior
from sbi_amd.utils.metrics import c2st
from tests.helpers import matched_pair
from tests.test_nsf_train_gpu import oracle_flat_grad

pytestmark = pytest.mark.gpu


class _OracleLogProb:
    """the oracle behind the two attributes log_prob_proposal_posterior_atomic needs"""

    def __init__(self, oracle, condition_shape):
        self.o, self.condition_shape = oracle, condition_shape

    def log_prob(self, input, condition):
        return self.o.log_prob(input, condition)


@pytest.mark.parametrize("combined", [False, True])
@pytest.mark.parametrize("cfg", [dict(D=4, C=7), dict(D=10, C=10), dict(D=4, C=7, hidden_features=100, num_transforms=3)],
                         ids=["D4-C7", "D10-C10", "D4-C7-hidden100"])
def test_fused_atomic_loss_and_grad_match_oracle_autograd(cfg, combined):
    oracle, est, theta_d, x_d = matched_pair(**cfg)
    B, A = 333, 10
    theta, x = theta_d[:B], x_d[:B]
    D = cfg["D"]
    prior_c = MultivariateNormal(torch.zeros(D), 0.1 * torch.eye(D))
    prior_g = MultivariateNormal(torch.zeros(D, device="cuda"), 0.1 * torch.eye(D, device="cuda"))
    masks = (torch.arange(B) % 2 == 0)[:, None]
    torch.manual_seed(5)
    choices = atomic.sample_contrasting_indices(B, A, "cpu")
    oracle.zero_grad()
    lpp = atomic.log_prob_proposal_posterior_atomic(_OracleLogProb(oracle, x[0].shape), prior_c, theta, x, masks, A,
                                                    combined, choices=choices)
    (-lpp).mean().backward()
    gref = oracle_flat_grad(oracle, est)

    stepper = FusedTrainStep(est, distributed=False)
    losses = stepper.atomic_loss_and_grad(theta.cuda(), x.cuda(), masks.cuda(), prior_g, A, combined,
                                          choices=choices.cuda())
    torch.cuda.synchronize()
    assert (losses.cpu() + lpp.detach()).abs().max() <= 2e-5 * (1 + lpp.detach().abs().max())
    got = stepper.grad.cpu()
    scale = gref.abs().max().item()
    rel = (got - gref).abs().max().item() / scale
    print(f"atomic grad: max|ref|={scale:.3e} rel={rel:.3e}")
    assert rel <= 3e-4


def test_two_round_npe_c_linear_gaussian_c2st():
    """Round 1 from the prior (MLE), round 2 from the round-1 posterior at x_o (atomic loss, 10 atoms)."""
    dim, n = 2, 1500
    torch.manual_seed(0)
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
    x_o = torch.zeros(1, dim)
    target = true_posterior_linear_gaussian_mvn_prior(x_o, shift, cov, torch.zeros(dim), torch.eye(dim)).sample((1000,))
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator=NSFConfig(), device="cuda", show_progress_bars=False)
    proposal = prior
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rnd in range(2):
            theta = proposal.sample((n,)).cpu()
            x = linear_gaussian(theta, shift, cov)
            inf.append_simulations(theta, x, proposal=proposal).train(training_batch_size=100, num_atoms=10)
            proposal = inf.build_posterior().set_default_x(x_o)
    assert inf._round == 1 and inf._data_round_index == [0, 1]
    samples = proposal.sample((1000,), show_progress_bars=False).cpu()
    score = c2st(samples, target).item()
    print(f"two-round NPE-C c2st={score:.3f} epochs={inf.summary['epochs_trained']}")
    assert 0.4 <= score <= 0.6


def test_device_atom_sampler_is_uniform_without_replacement():
    """csrc/atomic.hip: contrasting rows as npe_c.py:387-392 draws them -- num_atoms - 1 distinct rows != b, every row
    equally likely at every position -- and the atoms-major atom tensor built from them."""
    from sbi_amd import _lib

    lib = _lib.load()
    B, A, D = 50, 8, 3
    theta = torch.randn(B, D, device="cuda")
    counts = torch.zeros(A - 1, B, B, dtype=torch.long)          # [position, row, picked row]
    reps = 400
    for seed in range(1, reps + 1):
        ch = torch.empty(B, A - 1, dtype=torch.int64, device="cuda")
        atoms = torch.empty(A * B, D, device="cuda")
        rc = lib.sbi_amd_atomic_atoms(_lib.ptr(theta), B, A, D, seed * 7919, None, _lib.ptr(ch), _lib.ptr(atoms),
                                      _lib.current_stream(torch.device("cuda")))
        assert rc == 0
        c = ch.cpu()
        own = torch.arange(B)[:, None]
        assert ((c >= 0) & (c < B) & (c != own)).all()
        assert all(len(set(r.tolist())) == A - 1 for r in c)
        assert torch.equal(atoms.reshape(A, B, D)[0], theta) and torch.equal(atoms.reshape(A, B, D)[1:],
                                                                              theta[ch].transpose(0, 1))
        for pos in range(A - 1):
            counts[pos, torch.arange(B), c[:, pos]] += 1
    # every (position, row) histogram over the B - 1 other rows: expected reps / (B - 1) each
    exp = reps / (B - 1)
    off_diag = counts[:, ~torch.eye(B, dtype=torch.bool)].reshape(A - 1, B, B - 1).float()
    chi2 = ((off_diag - exp) ** 2 / exp).sum(-1)                   # ~ chi-square with B - 2 = 48 degrees of freedom
    assert chi2.mean() < 48 + 3 * (2 * 48 / (B * (A - 1))) ** 0.5 * 5 and chi2.max() < 110, (chi2.mean(), chi2.max())
    # totals per picked row, over everything: flat
    tot = counts.sum((0, 1)).float()
    assert (tot / tot.mean() - 1).abs().max() < 0.08


def test_fused_atomic_step_with_the_device_sampler_trains():
    oracle, est, theta_d, x_d = matched_pair(D=4, C=3)
    prior_g = MultivariateNormal(torch.zeros(4, device="cuda"), 0.1 * torch.eye(4, device="cuda"))
    B, A = 512, 10
    th, xx = theta_d[:B].cuda(), x_d[:B].cuda()
    masks = torch.zeros(B, 1, device="cuda")
    stepper = FusedTrainStep(est, distributed=False)
    torch.manual_seed(3)
    first = stepper.atomic_step(th, xx, masks, prior_g, A).mean().item()
    for _ in range(30):
        last = stepper.atomic_step(th, xx, masks, prior_g, A).mean().item()
    assert torch.isfinite(torch.tensor(last)) and last < first
    # same torch seed -> same contrasting sets -> same losses (the kernel's Philox key comes from torch's generator)
    _, est2, _, _ = matched_pair(D=4, C=3)
    st2 = FusedTrainStep(est2, distributed=False)
    torch.manual_seed(3)
    assert abs(st2.atomic_step(th, xx, masks, prior_g, A).mean().item() - first) < 1e-6
