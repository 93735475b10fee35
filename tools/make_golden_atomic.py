#!/usr/bin/env python
"""Generate tests/golden/atomic_reference.pt: the REAL `NPE_C._log_prob_proposal_posterior_atomic`
(sbi/inference/trainers/npe/npe_c.py:356-440) evaluated on a small analytic conditional density, with the
contrasting-set choices it drew (torch.multinomial) recorded.  Build container only."""

import os
import sys
import types

import torch



class GaussianRegression(torch.nn.Module):
    """q(theta | x) = N(theta; A x + b, diag(s^2)): the estimator interface the loss needs."""

    def __init__(self, D, C):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.A = torch.nn.Parameter(torch.randn(D, C, generator=g) * 0.5)
        self.b = torch.nn.Parameter(torch.randn(D, generator=g) * 0.1)
        self.log_s = torch.nn.Parameter(torch.randn(D, generator=g) * 0.2)
        self.input_shape, self.condition_shape = torch.Size([D]), torch.Size([C])

    def log_prob(self, input, condition):          # input (S, B, D), condition (B, C)
        mu = condition @ self.A.T + self.b
        z = (input - mu) / self.log_s.exp()
        return -0.5 * (z**2).sum(-1) - self.log_s.sum() - 0.5 * input.shape[-1] * torch.log(torch.tensor(2 * torch.pi))


def main():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_golden  # installs the third-party stubs and puts /root/reference on sys.path

    for mod in ["matplotlib", "matplotlib.pyplot", "matplotlib.axes", "matplotlib.figure", "joblib"]:
        try:
            __import__(mod)
        except Exception:
            make_golden.stub(mod)
    from sbi.inference.trainers.npe.npe_c import NPE_C

    D, C, B, A = 3, 4, 30, 7
    torch.manual_seed(21)
    theta, x = torch.randn(B, D), torch.randn(B, C)
    masks = (torch.arange(B) % 3 == 0)[:, None]
    prior = torch.distributions.MultivariateNormal(torch.zeros(D), 2.0 * torch.eye(D))
    est = GaussianRegression(D, C)
    out = {}
    for combined in (False, True):
        self = types.SimpleNamespace(_num_atoms=A, _prior=prior, _neural_net=est, _use_combined_loss=combined)
        recorded = {}
        real_multinomial = torch.multinomial

        def rec(*a, **k):
            r = real_multinomial(*a, **k)
            recorded["choices"] = r.clone()
            return r

        torch.multinomial = rec
        try:
            torch.manual_seed(22)
            lpp = NPE_C._log_prob_proposal_posterior_atomic(self, theta, x, masks)
        finally:
            torch.multinomial = real_multinomial
        grads = torch.autograd.grad(-lpp.sum(), list(est.parameters()))
        out[combined] = dict(choices=recorded["choices"], lpp=lpp.detach(), grads=[g.clone() for g in grads])
    g = dict(D=D, C=C, B=B, A=A, theta=theta, x=x, masks=masks, state=est.state_dict(), out=out)
    torch.save(g, os.path.join(make_golden.OUT, "atomic_reference.pt"))
    print("wrote atomic_reference.pt", out[False]["lpp"][:3])


if __name__ == "__main__":
    main()
