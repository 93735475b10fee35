"""SBI_AMD_TIMELINE=1 python tools/diag/coop_timeline.py [batch]: cycle stamps of the cooperative forward kernel."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd.inference.trainers.fused import FusedTrainStep
from sbi_amd.neural_nets.net_builders.flow import build_nsf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
theta = torch.randn(20000, 10); x = theta + 0.3 * torch.randn(20000, 10)
est = build_nsf(theta, x).cuda()
st = FusedTrainStep(est)
tb, xb = theta[:B].cuda(), x[:B].cuda()
for _ in range(30):
    st.step(tb, xb)
torch.cuda.synchronize()
