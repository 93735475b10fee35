"""Measured distances behind the asserted bars of tests/test_nsf_parity_gpu.py (inverse_transform vs oracle, round
trips at 2 048 and 65 536 rows): run on the GPU box, prints one line per quantity."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.helpers import make_inputs, matched_pair

oracle, est, _, _ = matched_pair(D=10, C=10)
theta, x = make_inputs(2048, 10, 10)
with torch.no_grad():
    ref = oracle.inverse_transform(theta, x)
noise = est.inverse_transform(theta.cuda(), x.cuda())
print("inverse_transform vs oracle, 2048 rows: max", float((noise.cpu() - ref).abs().max()))
back = est.sample_from_noise(noise, x.cuda())
print("sample(inverse(theta)) - theta, 2048 rows (tails included): max", float((back.cpu() - theta).abs().max()))
n = 65536
g = torch.Generator().manual_seed(9)
noise = torch.randn(n, 10, generator=g).cuda()
xx = (torch.randn(n, 10, generator=g) * 0.45).cuda()
th, ld = est.sample_from_noise(noise, xx, with_logabsdet=True)
back = est.inverse_transform(th, xx)
print("inverse(sample(noise)) - noise, 65536 rows: max", float((back - noise).abs().max()))
lp = est.log_prob(th, xx)[0]
base = -0.5 * (noise**2).sum(1) - est.net._log_z.to(noise.device).float()
d = (lp - (base - ld)).abs()
print("log_prob(sample) - (base - logabsdet), 65536 rows: max", float(d.max()), "99.9 %", float(d.quantile(0.999)))
