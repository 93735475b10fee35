"""DirectPosterior.sample of 10^6 draws for one x_o under a Gaussian prior (acceptance 1) and two box priors
(acceptance < 1): wall ms per call.  usage: python tools/diag/sample_timing.py [draws]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torch.distributions import Independent, Normal
from bench import make_data, build_estimator
from sbi_amd.inference.posteriors.direct_posterior import DirectPosterior
from sbi_amd.utils.torchutils import BoxUniform

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
theta, x = make_data(65536, dev)
est = build_estimator(*make_data(65536, "cpu"), dev)
D = theta.shape[1]
x_o = x[:1].clone()
priors = {"gaussian": Independent(Normal(torch.zeros(D, device=dev), (0.1**0.5) * torch.ones(D, device=dev)), 1),
          "box(-1,1)": BoxUniform(-torch.ones(D, device=dev), torch.ones(D, device=dev)),
          "box(-0.5,0.5)": BoxUniform(-0.5 * torch.ones(D, device=dev), 0.5 * torch.ones(D, device=dev))}
for name, prior in priors.items():
    post = DirectPosterior(est, prior, device=dev)
    for _ in range(2):
        s = post.sample((nd,), x=x_o, max_sampling_batch_size=nd, show_progress_bars=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        s = post.sample((nd,), x=x_o, max_sampling_batch_size=nd, show_progress_bars=False)
    torch.cuda.synchronize()
    print(f"{name:14s} {1e3 * (time.perf_counter() - t0) / K:7.3f} ms per {nd} draws  (shape {tuple(s.shape)})")
