"""Adaptive Dormand-Prince 5(4) integrator for the probability-flow ODE of a vector-field estimator.

Plays the role of zuko's ``odeint`` inside ``FreeFormJacobianTransform`` as sbi configures it
(sbi/samplers/ode_solvers/zuko_ode.py:19-124: atol 1e-6, rtol 1e-5, integrate between t_max and t_min).
zuko is a third-party dependency that is absent here, so the step-size controller is the textbook one
(Hairer, Norsett, Wanner, Solving ODEs I, II.4) rather than a restatement of zuko's: PARITY UNPINNED at that
boundary -- the ODE solution itself is unique, and tests compare against a tight-tolerance solve of the oracle's
vector field.

All state stays on the device, the step-size controller included (csrc/ode.hip); every right-hand-side evaluation
is one launch of the HIP velocity kernel over the whole batch, every stage combination one fused pass, and the host
never blocks on the attempt it has just enqueued.
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
    """Integrate dy/dt = f(t, y) from t0 to t1 (either direction); ``f`` takes a 1-element time tensor.

    fp32 state on a ROCm device: the device-resident stepper (csrc/ode.hip).  Anything else (the CPU tests of the
    host logic, fp64 states): the same method written with torch ops and a host-side controller."""
    if float(t1) == float(t0):
        return y0.clone()
    if y0.is_cuda and y0.dtype == torch.float32:
        return _odeint_device(f, y0, float(t0), float(t1), atol, rtol, max_steps, first_step)
    return _odeint_host(f, y0, float(t0), float(t1), atol, rtol, max_steps, first_step)


def _odeint_device(f, y0: Tensor, t0: float, t1: float, atol: float, rtol: float, max_steps: int,
                   first_step: float, use_graph: bool = False) -> Tensor:
    """Time, step size, the accept / reject decision, the controller and the FSAL hand-over live in a 32-float state
    block on the device (include/sbi_amd_fmpe.h); one attempt is a FIXED sequence of launches -- 6 x (stage
    combination, velocity) and the finish pair -- with no host decision inside.  The host does not wait for the
    attempt it has just enqueued: the two flags it needs ("t has reached t1", "the attempt now in flight reaches t1
    if accepted") are copied to pinned memory after every attempt and read ONE ATTEMPT LATE; only when the second flag
    says the attempt in flight may be the last does the host wait for it, so no attempt is enqueued behind the end.

    ``use_graph``: capture the attempt once into a HIP graph and replay it.  Measured on MI355X
    (tools/diag/ode_timing.py): capturing costs about as much as 10 eager attempts and a solve of the trained flows
    takes 8-25, so it is off by default -- it pays only for callers that integrate for hundreds of attempts."""
    from ctypes import c_void_p

    from sbi_amd import _lib

    lib = _lib.load()
    dev = _lib.require_device(y0)
    y = y0.contiguous().clone()
    n = y.numel()
    state = torch.zeros(32, dtype=torch.float32, device=dev)
    scratch = torch.empty(256, dtype=torch.float64, device=dev)
    y_stage = [torch.empty_like(y) for _ in range(2)]

    def time_slot(i: int) -> Tensor:      # 1-element views the velocity kernel reads its time from
        return state[i : i + 1]

    def own(v: Tensor) -> Tensor:
        if v.shape != y.shape or v.dtype != torch.float32 or not v.is_contiguous():
            v = v.to(torch.float32).reshape(y.shape).contiguous()
        return v

    def attempt(k1: Tensor) -> None:
        stream = _lib.current_stream(dev)
        ks = [k1]
        yi = y
        for i in range(1, 7):
            yi = y_stage[i & 1]
            kp = (c_void_p * 7)(*[_lib.ptr(k) for k in ks], *([None] * (7 - len(ks))))
            _lib.check(lib.sbi_amd_dopri5_stage(_lib.ptr(y), kp, i, _lib.ptr(state), _lib.ptr(yi), n, stream),
                       "dopri5_stage")
            k = own(f(time_slot(16 + i), yi))
            # (the seven stage derivatives are read together at the end of the attempt: a right-hand side that hands
            # back its input or one persistent output buffer gets a private copy)
            if k.data_ptr() == yi.data_ptr() or any(k.data_ptr() == o.data_ptr() for o in ks):
                k = k.clone()
            ks.append(k)
        kp = (c_void_p * 7)(*[_lib.ptr(k) for k in ks])
        _lib.check(lib.sbi_amd_dopri5_finish(_lib.ptr(y), _lib.ptr(yi), kp, _lib.ptr(state), _lib.ptr(scratch), n,
                                             stream), "dopri5_finish")

    with torch.cuda.device(dev):
        _lib.check(lib.sbi_amd_dopri5_init(_lib.ptr(state), t0, t1, float(first_step), float(atol), float(rtol),
                                           _lib.current_stream(dev)), "dopri5_init")
        k1 = own(f(time_slot(23), y)).clone()      # ours: the finish kernel overwrites it on acceptance (FSAL)
        graph = None
        flags = []
        maybe_last = float(first_step) >= abs(t1 - t0)      # does the attempt about to be enqueued reach t1?
        for it in range(max_steps):
            if graph is not None:
                graph.replay()
            else:
                attempt(k1)
                if use_graph and it == 0:
                    # (the eager attempt above doubled as the warm-up: every lazy initialisation inside f is done)
                    try:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):
                            attempt(k1)
                        graph = g
                    except Exception:   # noqa: BLE001 -- not capturable: stay eager
                        graph = None
                        use_graph = False
                        torch.cuda.synchronize(dev)
            host = torch.empty(5, dtype=torch.float32, pin_memory=True)
            host.copy_(state[24:29], non_blocking=True)      # [0] finished, [4] next attempt may be the last
            ev = torch.cuda.Event()
            ev.record()
            # flags of the PREVIOUS attempt (read one attempt late) say whether the attempt just enqueued may be the
            # last one; if so -- or if that was known already -- wait for it now instead of queueing behind the end
            wait_now = maybe_last
            if not wait_now and flags:
                prev, pev = flags.pop(0)
                pev.synchronize()
                if float(prev[0]) != 0.0:
                    return y            # (the attempt enqueued after `prev` ran with h = 0: y is unchanged)
                wait_now = float(prev[4]) != 0.0
            if wait_now:
                ev.synchronize()
                if float(host[0]) != 0.0:
                    return y
                maybe_last = float(host[4]) != 0.0          # rejected (or not quite there): how about the next one
                flags = []
            else:
                maybe_last = False
                flags = [(host, ev)]
    raise RuntimeError("odeint_dopri5: max_steps exceeded")


def _odeint_host(f, y0: Tensor, t0: float, t1: float, atol: float, rtol: float, max_steps: int,
                 first_step: float) -> Tensor:
    direction = 1.0 if t1 >= t0 else -1.0
    span = abs(t1 - t0)
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
        ratio = float(torch.sqrt(torch.mean((err / scale) ** 2)))
        if ratio <= 1.0:
            t += hs
            y = y5
            k1 = ks[6]
        factor = 0.9 * ratio ** (-0.2) if ratio > 0.0 else 5.0
        h *= min(5.0, max(0.2, factor))
    raise RuntimeError("odeint_dopri5: max_steps exceeded")
