"""hidden_features 50 / 100 / 128: device time of log_prob, of the sampling direction and of the fused training step at a few
batch sizes (the wide cooperative kernels take every call above hidden 64).  usage: python tools/diag/wide_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd.inference.trainers.fused import FusedTrainStep
from sbi_amd.neural_nets.net_builders.flow import build_nsf


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
N = 100_000
theta = torch.randn(N, 10)
x = theta + 0.3 * torch.randn(N, 10)
for H in (50, 100, 128):
    est = build_nsf(theta, x, hidden_features=H).cuda()
    st = FusedTrainStep(est)
    for B in (200, 8192, 65536):
        tb, xb = theta[:B].cuda().contiguous(), x[:B].cuda().contiguous()
        noise = torch.randn(B, 10, device="cuda")
        reps = 50 if B < 65536 else 10
        t_lp = timed(lambda: est.log_prob(tb, xb), reps)
        t_sm = timed(lambda: est.sample_from_noise(noise, xb), reps)
        t_tr = timed(lambda: st.step(tb, xb), reps)
        print(f"hidden {H:3d} batch {B:6d}: log_prob {t_lp:.3f} ms, sample_from_noise {t_sm:.3f} ms, train step {t_tr:.3f} ms",
              flush=True)
