import torch, time, sys
sys.path.insert(0, "/root/repo")
from sbi_amd.inference.trainers.fused import FusedFMPEStep, FusedTrainStep
from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator
for D in (3, 50):
    th = torch.randn(4096, D); x = th + 0.3 * torch.randn(4096, D)
    fm = build_flow_matching_estimator(th, x).cuda()
    st = FusedFMPEStep(fm)
    for B in (200, 1000, 4096):
        t, xx = th[:B].cuda(), x[:B].cuda()
        for _ in range(20): st.step(t, xx)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): st.step(t, xx)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        # device time only
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): st.step(t, xx)
        e1.record(); torch.cuda.synchronize()
        print(f"FMPE D={D} B={B}: wall {dt*1e6:.0f} us/step, device {e0.elapsed_time(e1)/200*1e3:.0f} us/step")
