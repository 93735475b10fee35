"""Per-phase shader-cycle timeline of one row wave and one grad wave of the backward kernel
(first tile of workgroup 0, transform 0).  Run with SBI_AMD_TIMELINE=1 on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SBI_AMD_TIMELINE"] = "1"
os.environ.setdefault("SBI_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sbi_amd", "libsbi_amd_nsf_debug.so"))   # python -m sbi_amd._build --debug
from bench import make_data, build_estimator
from sbi_amd.inference.trainers.fused import FusedTrainStep
dev = torch.device("cuda:0")
theta, x = make_data(65536, dev)
est = build_estimator(*make_data(65536, "cpu"), dev)
st = FusedTrainStep(est)
for _ in range(3): st.step(theta, x)
torch.cuda.synchronize()
ts = st.workspace[-2048:].view(torch.int64).cpu().reshape(-1, 64)
for w, name in ((0, "row wave 0"), (4, "grad wave 0")):
    t = ts[w]
    pts = [(i, int(t[i])) for i in range(64) if int(t[i]) != 0]
    pts.sort(key=lambda a: a[1])
    if not pts:
        print(name, 'no timestamps'); continue
    t0 = pts[0][1]
    print(name, "total cycles", pts[-1][1] - t0)
    prev = t0
    for i, v in pts:
        print(f"  TS{i:2d} +{v - prev:7d}  @{v - t0:7d}")
        prev = v

# ---- forward kernel: timestamps land in the noise_out buffer (debug only)
from sbi_amd.neural_nets.estimators.nsf_flow import _log_prob_call
with torch.no_grad():
    for _ in range(2):
        lp, noise = _log_prob_call(est.net, theta, x, want_noise=True)
        noise.zero_() if _ == 0 else None
torch.cuda.synchronize()
ts = noise.reshape(-1)[:4096].view(torch.int64).cpu().reshape(-1, 64)
for w in (0, 4):
    t = ts[w]
    pts = sorted([(i, int(t[i])) for i in range(64) if 0 < int(t[i]) < 2**62], key=lambda a: a[1])
    if not pts:
        print("fwd wave", w, "no timestamps"); continue
    t0 = pts[0][1]
    print("forward wave", w, "layer 1 total cycles", pts[-1][1] - t0)
    prev = t0
    for i, v in pts:
        print(f"  TSF{i:2d} +{v - prev:7d}  @{v - t0:7d}")
        prev = v
