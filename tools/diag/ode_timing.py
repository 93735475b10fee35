"""FMPE posterior sampling: device-resident stepper (csrc/ode.hip) vs the host-controller loop of torch ops.
usage: python tools/diag/ode_timing.py [num_samples ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd.samplers.ode_solvers import dopri5
from tests.test_fmpe_gpu import make_pair

sizes = [int(a) for a in sys.argv[1:]] or [1000, 100_000]
oracle, est, theta, x, _, _ = make_pair(50, 50, n=256)
x_o = x[:1].cuda()
for n in sizes:
    eps = torch.randn(n, 50, device="cuda")
    f = lambda t, y: est.ode_fn(y, x_o, t)
    import functools
    for name, fn in (("host controller", dopri5._odeint_host), ("device stepper", dopri5._odeint_device),
                     ("device stepper, HIP graph", functools.partial(dopri5._odeint_device, use_graph=True))):
        try:
            fn(f, eps, 1.0, 0.0, 1e-6, 1e-5, 2_000, 0.05)
        except RuntimeError as e:
            print(name, "failed:", e)
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = fn(f, eps, 1.0, 0.0, 1e-6, 1e-5, 10_000, 0.05)
        torch.cuda.synchronize()
        print(f"n={n:7d} {name:22s}: {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms per solve", flush=True)
