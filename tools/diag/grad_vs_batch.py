"""Diagnostic: flat training gradient vs fp64 oracle autograd as the batch grows (1 tile per workgroup up to
16 384 rows, several beyond): separates fp32 accumulation growth from a multi-tile bug."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.helpers import matched_pair
from sbi_amd.inference.trainers.fused import FusedTrainStep

oracle, est, _, _ = matched_pair(D=10, C=10)
g = torch.Generator().manual_seed(2)
N = 65536
theta = torch.randn(N, 10, generator=g) * (0.1**0.5)
x = theta + (0.1**0.5) * torch.randn(N, 10, generator=g)
oracle.double()
named = dict(oracle.named_parameters())
stepper = FusedTrainStep(est, distributed=False)
for n in [int(a) for a in (sys.argv[1:] or [1024, 4096, 16384, 16448, 32768, 65536])]:
    oracle.zero_grad()
    for i in range(0, n, 16384):
        (oracle.loss(theta[i:min(n, i + 16384)].double(), x[i:min(n, i + 16384)].double()).sum() / n).backward()
    g64 = torch.zeros(est.net.flat_params.numel(), dtype=torch.float64)
    for key, off, cnt, _ in est.net._slices():
        g64[off:off + cnt] = named["net." + key].grad.reshape(-1)
    stepper._workspace(n).fill_(float("nan"))
    stepper.loss_and_grad(theta[:n].cuda(), x[:n].cuda())
    got = stepper.grad.cpu().double()
    scale = g64.abs().max().item()
    worst = {}
    for key, off, cnt, _ in est.net._slices():
        e = (got[off:off + cnt] - g64[off:off + cnt]).abs().max().item() / scale
        short = key.split("_transforms.")[1]
        t, rest = short.split(".", 1)
        kind = rest.replace("transform_net.", "")
        worst.setdefault(kind, (0.0, None))
        if e > worst[kind][0]:
            worst[kind] = (e, t)
    tot = (got - g64).abs().max().item() / scale
    print(f"n={n}: rel err {tot:.3e} (scale {scale:.3e})")
    for k, (e, t) in sorted(worst.items(), key=lambda kv: -kv[1][0])[:6]:
        print(f"    {k:45s} {e:.3e} (transform index {t})")
