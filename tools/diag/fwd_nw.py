"""Diagnostic: training forward with 8 waves per workgroup (n >= 32768) vs 4 (SBI_AMD_ABLATE=1024): stash contents
and resulting gradient."""
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
    ws = stepper._workspace(n)
    ws.fill_(0.0)
    losses = stepper.loss_and_grad(theta.cuda(), x.cuda())
    torch.save({"grad": stepper.grad.cpu(), "ws": ws.cpu(), "loss": losses.cpu()}, sys.argv[3])
else:
    import torch
    n = 32768
    outs = []
    for abl in ("0", "1024"):
        f = f"/tmp/w_{abl}.pt"
        subprocess.check_call([sys.executable, __file__, "child", str(n), f], env=dict(os.environ, SBI_AMD_ABLATE=abl))
        outs.append(torch.load(f))
    a, b = outs
    print("grad diff rel", (a["grad"] - b["grad"]).abs().max().item() / a["grad"].abs().max().item())
    print("loss diff", (a["loss"] - b["loss"]).abs().max().item())
    wa, wb = a["ws"], b["ws"]
    T, D = 5, 10
    o_stash = 0; o_noise = T * n * D; o_logp = o_noise + n * D; o_gza = o_logp + n; o_gzb = o_gza + n * D
    print("z stash diff", (wa[:o_noise] - wb[:o_noise]).abs().max().item())
    print("noise diff", (wa[o_noise:o_logp] - wb[o_noise:o_logp]).abs().max().item())
    o_part = (o_gzb + n * D + 3) // 4 * 4
    d = (wa - wb).abs()
    nz = d.nonzero().flatten()
    print("first differing workspace offsets", nz[:10].tolist(), "count", nz.numel(), "of", d.numel(), "o_part", o_part)
    big = d.argmax().item()
    print("largest diff at", big, d[big].item(), wa[big].item(), wb[big].item())
