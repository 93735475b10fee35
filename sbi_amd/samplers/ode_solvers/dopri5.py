"""Adaptive Dormand-Prince 5(4) integrator for the probability-flow ODE of a vector-field estimator.

Plays the role of zuko's ``odeint`` inside ``FreeFormJacobianTransform`` as sbi configures it
(sbi/samplers/ode_solvers/zuko_ode.py:19-124: atol 1e-6, rtol 1e-5, integrate between t_max and t_min).
zuko is a third-party dependency that is absent here, so the step-size controller is the textbook one
(Hairer, Norsett, Wanner, Solving ODEs I, II.4) rather than a restatement of zuko's: PARITY UNPINNED at that
boundary -- the ODE solution itself is unique, and tests compare against a tight-tolerance solve of the oracle's
vector field.

All state stays on the device; every right-hand-side evaluation is one launch of the HIP velocity kernel over
the whole batch, and the controller reads one scalar per attempted step.
"""

from __future__ import annotations

from typing import Callable

import torch
from torch import Tensor

# Butcher tableau (Dormand & Prince 1980)
_C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_A = (
    (),
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
    (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)
_B5 = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)
_B4 = (5179 / 57600, 0.0, 7571 / 16695, 393 / 640, -92097 / 339200, 187 / 2100, 1 / 40)


@torch.no_grad()
def odeint_dopri5(f: Callable[[Tensor, Tensor], Tensor], y0: Tensor, t0: float, t1: float, atol: float = 1e-6,
                  rtol: float = 1e-5, max_steps: int = 10_000, first_step: float = 0.05) -> Tensor:
    """Integrate dy/dt = f(t, y) from t0 to t1 (either direction); ``f`` takes a 1-element time tensor."""
    direction = 1.0 if t1 >= t0 else -1.0
    span = abs(t1 - t0)
    if span == 0.0:
        return y0.clone()
    y = y0.clone()
    t = float(t0)
    h = min(first_step, span)
    tt = torch.empty(1, dtype=y.dtype, device=y.device)

    def rhs(time: float, state: Tensor) -> Tensor:
        tt.fill_(time)
        return f(tt, state)

    k1 = rhs(t, y)
    for _ in range(max_steps):
        remaining = abs(t1 - t)
        if remaining <= 1e-12 * max(1.0, span):
            return y
        h = min(h, remaining)
        hs = direction * h
        ks = [k1]
        for i in range(1, 7):
            yi = y.clone()
            for a, k in zip(_A[i], ks):
                if a != 0.0:
                    yi.add_(k, alpha=hs * a)
            ks.append(rhs(t + hs * _C[i], yi))
        y5 = yi                                   # stage 7 is evaluated AT the 5th-order solution (FSAL)
        err = torch.zeros_like(y)
        for b5, b4, k in zip(_B5, _B4, ks):
            if b5 != b4:
                err.add_(k, alpha=hs * (b5 - b4))
        scale = atol + rtol * torch.maximum(y.abs(), y5.abs())
        ratio = float(torch.sqrt(torch.mean((err / scale) ** 2)))   # the one host read of the step
        if ratio <= 1.0:
            t += hs
            y = y5
            k1 = ks[6]
        factor = 0.9 * ratio ** (-0.2) if ratio > 0.0 else 5.0
        h *= min(5.0, max(0.2, factor))
    raise RuntimeError("odeint_dopri5: max_steps exceeded")
