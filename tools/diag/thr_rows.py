"""Device time of the THROUGHPUT kernels' log_prob / sampling direction at row counts that do not fill whole rounds of 256
workgroups (the workgroup size is chosen per call: csrc/nsf_plan.cpp::nsf_plan_for_rows)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd import _lib
from sbi_amd.neural_nets.net_builders.flow import build_nsf
torch.manual_seed(0)
theta = torch.randn(70000, 10); x = theta + 0.3 * torch.randn(70000, 10)
est = build_nsf(theta, x).cuda()
theta, x = theta.cuda(), x.cuda()
_lib.load().sbi_amd_nsf_set_coop_max_rows(0)
def dev_ms(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for B in (4096, 8192, 10000, 12288, 16384, 20000, 24576, 32768, 40000, 65536):
    tb, xb = theta[:B].contiguous(), x[:B].contiguous()
    nz = torch.randn(B, 10, device="cuda")
    with torch.no_grad():
        print(f"rows {B:6d}: log_prob {dev_ms(lambda: est.log_prob(tb, xb)):.3f} ms, sample {dev_ms(lambda: est.sample_from_noise(nz, xb)):.3f} ms", flush=True)
