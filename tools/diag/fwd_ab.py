"""A/B of kernel-library builds on one box: paired log_prob (65 536 rows), log_prob with one x_o, 10^6 draws, the fused
training step, and the distances the parity gates bound.  usage: SBI_AMD_LIB=... python tools/diag/fwd_ab.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import make_data, build_estimator
from sbi_amd.inference.trainers.fused import FusedTrainStep

dev = torch.device("cuda:0")
n = 65536
theta, x = make_data(n, dev)
est = build_estimator(*make_data(n, "cpu"), dev)


def timed(fn, k=50, w=10):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


with torch.no_grad():
    t_lp = timed(lambda: est.log_prob(theta, x))
    t_bx = timed(lambda: est.log_prob(theta, x[:1]))
    noise = torch.randn(1_000_000, theta.shape[1], device=dev)
    t_s = timed(lambda: est.sample_from_noise(noise, x[:1]), k=20, w=3)
st = FusedTrainStep(est)
t_tr = timed(lambda: st.step(theta, x), k=100, w=10)
print(f"log_prob paired {t_lp:.4f} ms | one x_o {t_bx:.4f} ms | 1e6 draws {t_s:.3f} ms | fused step {t_tr:.4f} ms")

# accuracy against the oracle's fp64 evaluation (16 384 in-distribution rows of the default shape)
from tests.helpers import matched_pair
oracle, est2, th_d, x_d = matched_pair(D=10, C=10, n=16384)
with torch.no_grad():
    ref32 = oracle.log_prob(th_d, x_d)[0]
    ref64 = oracle.double().log_prob(th_d.double(), x_d.double())[0]
    oracle.float()
    got = est2.log_prob(th_d.cuda(), x_d.cuda())[0].cpu().double()
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(16384, 10, generator=g)
    s64, ld64 = oracle.double().sample_from_noise(noise.double(), x_d.double())
    oracle.float()
    s, ld = est2.sample_from_noise(noise.cuda(), x_d.cuda(), with_logabsdet=True)
d = (got - ref64).abs()
d32 = (ref32.double() - ref64).abs()
print(f"log_prob vs f64: rms {d.pow(2).mean().sqrt():.3e} max {d.max():.3e} beyond 1e-5 {float((d > 1e-5).double().mean()):.4%}"
      f" | eager fp32 oracle: rms {d32.pow(2).mean().sqrt():.3e} max {d32.max():.3e} beyond 1e-5 {float((d32 > 1e-5).double().mean()):.4%}")
ds = (s.cpu().double() - s64).abs()
dl = (ld.cpu().double() - ld64).abs()
print(f"sample vs f64: theta rms {ds.pow(2).mean().sqrt():.3e} max {ds.max():.3e} | logabsdet rms {dl.pow(2).mean().sqrt():.3e} max {dl.max():.3e}")
