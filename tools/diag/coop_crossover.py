"""Where do the cooperative small-batch kernels stop paying?  Device time of log_prob, of the sampling direction and of
the fused training step at several batch sizes, once per kernel family (sbi_amd_nsf_set_coop_max_rows)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd import _lib
from sbi_amd.inference.trainers.fused import FusedTrainStep
from sbi_amd.neural_nets.net_builders.flow import build_nsf

torch.manual_seed(0)
N = 70000
theta = torch.randn(N, 10); x = theta + 0.3 * torch.randn(N, 10)
est = build_nsf(theta, x).cuda()
theta, x = theta.cuda(), x.cuda()
lib = _lib.load()


def dev_ms(fn, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for B in [int(a) for a in sys.argv[1:]] or [200, 1024, 4096, 8192, 12288, 16384, 24576, 32768, 65536]:
    tb, xb = theta[:B].contiguous(), x[:B].contiguous()
    out = []
    for fam, rows in (("coop", 1 << 40), ("thr", 0)):
        lib.sbi_amd_nsf_set_coop_max_rows(rows)
        st = FusedTrainStep(est)
        nz = torch.randn(B, 10, device="cuda")
        with torch.no_grad():
            lp = dev_ms(lambda: est.log_prob(tb, xb))
            sm = dev_ms(lambda: est.sample_from_noise(nz, xb))
        tr = dev_ms(lambda: st.step(tb, xb))
        out.append(f"{fam}: log_prob {lp:.3f} ms, sample {sm:.3f} ms, train step {tr:.3f} ms")
    print(f"batch {B:6d} | " + " | ".join(out), flush=True)
