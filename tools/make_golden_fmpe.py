#!/usr/bin/env python
"""Generate tests/golden/fmpe_reference.pt from the REAL sbi classes (build container only):
`build_vector_field_estimator` -> FlowMatchingEstimator + VectorFieldMLP
(sbi/neural_nets/net_builders/vector_field_nets.py:136-339, 610-719; estimators/flowmatching_estimator.py).
Stored per case: the state_dict (all parameters perturbed so the zero-initialised output layer and the unit
LayerNorm gains are exercised), inputs, the noise the loss drew, per-row losses, d mean-loss / d parameters,
the velocity `forward()` returns at a few (theta_t, t) for one observation, and the exact Jacobian trace of
`ode_fn` there (the integrand of the log-density along the probability-flow ODE)."""

import os
import sys

import torch


def main():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_golden  # third-party stubs + /root/reference on sys.path

    for mod in ["matplotlib", "matplotlib.pyplot", "matplotlib.axes", "matplotlib.figure", "joblib"]:
        try:
            __import__(mod)
        except Exception:
            make_golden.stub(mod)
    from sbi.neural_nets.net_builders.vector_field_nets import build_vector_field_estimator

    cases = {}
    for name, (D, C, kw) in {
        "default_D5_C3": (5, 3, {}),
        "H48_L2_D3_C4": (3, 4, dict(hidden_features=48, num_layers=2)),
    }.items():
        torch.manual_seed(7)
        theta = torch.randn(300, D) * torch.linspace(0.5, 3.0, D) + torch.linspace(-2.0, 2.0, D)
        x = theta[:, :1] * torch.ones(1, C) + torch.randn(300, C) * 0.3 + 1.5
        est = build_vector_field_estimator(theta, x, **kw)
        with torch.no_grad():
            for p in est.parameters():
                p.add_(0.05 * torch.randn_like(p))
        n = 64
        times = torch.rand(n)
        torch.manual_seed(11)
        noise = torch.randn_like(theta[:n])
        torch.manual_seed(11)      # the loss draws theta_1 = randn_like(input) as its only random call
        losses = est.loss(theta[:n], x[:n], times=times)
        est.zero_grad()
        losses.mean().backward()
        grads = {k: p.grad.clone() for k, p in est.named_parameters()}
        tq = torch.tensor([0.0, 0.05, 0.3, 0.5, 0.77, 1.0]).repeat_interleave(4)
        theta_q = torch.randn(tq.shape[0], D) * 1.5
        with torch.no_grad():
            vel = est(theta_q, x[:1], tq)
        # exact trace of d ode_fn / d input: what zuko's FreeFormJacobianTransform(exact=True) integrates for
        # VectorFieldPosterior.log_prob (zuko itself is not installed here: the trace is taken with plain autograd,
        # one reverse pass per theta dim, on the real estimator's ode_fn)
        thq = theta_q.clone().requires_grad_(True)
        vq = est.ode_fn(thq, x[:1], tq)
        div = torch.zeros(tq.shape[0])
        for f in range(D):
            (gf,) = torch.autograd.grad(vq[:, f].sum(), thq, retain_graph=True)
            div += gf[:, f]
        cases[name] = dict(div=div.detach(), D=D, C=C, kw=kw, state=est.state_dict(), theta=theta[:n].clone(), x=x[:n].clone(),
                           times=times, noise=noise, losses=losses.detach(), grads=grads, tq=tq, theta_q=theta_q,
                           vel=vel)
        print(name, "loss", losses[:3].tolist(), "params", sum(p.numel() for p in est.parameters()))
    torch.save(cases, os.path.join(make_golden.OUT, "fmpe_reference.pt"))


if __name__ == "__main__":
    main()
