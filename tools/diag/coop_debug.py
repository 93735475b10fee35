"""Step-by-step check of the cooperative kernels (synchronising after every call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd import _lib   # noqa: E402
from tests.helpers import matched_pair   # noqa: E402

oracle, est, theta, x = matched_pair(D=10, C=10, n=3000)
lib = _lib.load()
for n in [200, 777, 16, 1, 2048]:
    for want_noise in (False, True):
        th, xx = theta[:n].cuda(), x[:n].cuda()
        print("log_prob n", n, "noise", want_noise, flush=True)
        lp, nz = est._kernel_log_prob(th, xx, want_noise)
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = oracle.log_prob(theta[:n], x[:n])[0]
        print("   max err", (lp.cpu() - ref).abs().max().item(), flush=True)
print("done")
from sbi_amd.neural_nets.estimators.nsf_flow import train_forward, train_workspace   # noqa: E402

for n in [200, 777, 333, 16, 1, 4097]:
    th, xx = theta[:n].cuda(), x[:n].cuda()
    print("train_forward n", n, flush=True)
    ws = train_workspace(est.net, n, "cuda")
    print("   ws floats", ws.numel(), flush=True)
    lp = train_forward(est.net, th, xx, ws)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = oracle.log_prob(theta[:n], x[:n])[0]
    print("   max err", (lp.cpu() - ref).abs().max().item(), flush=True)
print("autograd log_prob", flush=True)
lp = est.log_prob(theta[:777].cuda(), x[:777].cuda())
torch.cuda.synchronize()
print("ok", lp.shape, flush=True)
print("done2")
