"""Which rows of the 65 536-row training batch disagree with the fp64 oracle in d loss / d theta, and how far are
their spline inputs from a knot?  (evidence for tests/test_parity_full_size_gpu.py; run on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd.neural_nets.estimators.nsf_flow import loss_fwd_bwd, train_workspace   # noqa: E402
from tests.helpers import matched_pair, spline_knot_distances   # noqa: E402

N, CHUNK = 65536, 16384
oracle, est, _, _ = matched_pair(D=10, C=10)
g = torch.Generator().manual_seed(2)
theta = torch.randn(N, 10, generator=g) * (0.1**0.5)
x = theta + (0.1**0.5) * torch.randn(N, 10, generator=g)
oracle.double()
gth = []
for i in range(0, N, CHUNK):
    th = theta[i:i + CHUNK].double().requires_grad_(True)
    oracle.loss(th, x[i:i + CHUNK].double()).sum().backward()
    gth.append(th.grad)
oracle.float()
gth64 = torch.cat(gth)
grad = torch.empty_like(est.net.flat_params.data)
_, gth_h = loss_fwd_bwd(est.net, theta.cuda(), x.cuda(), None, 1.0 / N, grad, want_grad_theta=True,
                        workspace=train_workspace(est.net, N, "cuda"))
gth_h = gth_h.cpu() * N
row_err = (gth_h.double() - gth64).abs().max(dim=1).values / gth64.abs().max().item()
order = row_err.argsort(descending=True)[:40]
dist = spline_knot_distances(oracle, theta[order], x[order]).min(dim=1).values
for r, e, d in zip(order.tolist(), row_err[order].tolist(), dist.tolist()):
    print(f"row {r:6d}  rel d loss/d theta err {e:.3e}  closest spline input to a knot: {d:10.2f} fp32 spacings at B")
for thr in (1e-3, 1e-4, 3e-5, 1e-5, 3e-6):
    print(f"rows with err > {thr:g}: {(row_err > thr).sum().item()}")
