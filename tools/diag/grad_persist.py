"""Diagnostic: does the backward kernel's multi-tile persistence change the gradient?  SBI_AMD_ABLATE=512 caps the
grid at 4 workgroups, so a 2 048-row batch runs 8 tiles per workgroup."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from tests.helpers import matched_pair
    from sbi_amd.inference.trainers.fused import FusedTrainStep
    oracle, est, _, _ = matched_pair(D=10, C=10)
    g = torch.Generator().manual_seed(2)
    n = int(sys.argv[2])
    theta = torch.randn(n, 10, generator=g) * (0.1**0.5)
    x = theta + (0.1**0.5) * torch.randn(n, 10, generator=g)
    stepper = FusedTrainStep(est, distributed=False)
    stepper._workspace(n).fill_(float("nan"))
    stepper.loss_and_grad(theta.cuda(), x.cuda())
    torch.save(stepper.grad.cpu(), sys.argv[3])
else:
    import torch
    for n in (2048, 24576, 32768):
        outs = []
        for abl in ("0", "512"):
            f = f"/tmp/g_{n}_{abl}.pt"
            env = dict(os.environ, SBI_AMD_ABLATE=abl)
            subprocess.check_call([sys.executable, __file__, "child", str(n), f], env=env)
            outs.append(torch.load(f))
        a, b = outs
        print(f"n={n}: max|grid-capped - normal| / max|g| = {(a - b).abs().max().item() / a.abs().max().item():.3e}")
