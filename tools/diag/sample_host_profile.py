"""Where the wall time of DirectPosterior.sample(10^6 draws, one x_o) goes beyond the sampling kernel: device time of
the call's kernels vs wall, and a cProfile of the host side.  usage: python tools/diag/sample_host_profile.py"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torch.distributions import Independent, Normal
from bench import make_data, build_estimator
from sbi_amd.inference.posteriors.direct_posterior import DirectPosterior

nd = 1_000_000
dev = torch.device("cuda:0")
theta, x = make_data(65536, dev)
est = build_estimator(*make_data(65536, "cpu"), dev)
D = theta.shape[1]
x_o = x[:1].clone()
prior = Independent(Normal(torch.zeros(D, device=dev), (0.1**0.5) * torch.ones(D, device=dev)), 1)
post = DirectPosterior(est, prior, device=dev)
call = lambda: post.sample((nd,), x=x_o, max_sampling_batch_size=nd, show_progress_bars=False)
for _ in range(3):
    call()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    call()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K * 1e3
noise = torch.randn(nd, D, device=dev)
for _ in range(3):
    est.sample_from_noise(noise, x_o)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    est.sample_from_noise(noise, x_o)
torch.cuda.synchronize()
kern = (time.perf_counter() - t0) / K * 1e3
t0 = time.perf_counter()
for _ in range(K):
    est.sample((nd,), x_o)
torch.cuda.synchronize()
smp = (time.perf_counter() - t0) / K * 1e3
print(f"posterior.sample {wall:.3f} ms | estimator.sample (randn + kernel) {smp:.3f} ms | sample_from_noise {kern:.3f} ms")
# host time until the sampling kernel is enqueued: time the call with the device idle, up to the first sync
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    call()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
