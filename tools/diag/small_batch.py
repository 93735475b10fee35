"""Where does a training step go at small batches?  Wall time per step of the host-driven loop (no syncs inside),
device time per step (events), and the same step replayed from a captured HIP graph (torch.cuda.CUDAGraph around the
ctypes launches).  usage: python tools/diag/small_batch.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd.inference.trainers.fused import FusedTrainStep
from sbi_amd.neural_nets.net_builders.flow import build_nsf

batches = [int(a) for a in sys.argv[1:]] or [200, 1024, 4096, 8192, 65536]
torch.manual_seed(0)
N = 100_000
theta = torch.randn(N, 10)
x = theta + 0.3 * torch.randn(N, 10)
est = build_nsf(theta, x).cuda()
theta, x = theta.cuda(), x.cuda()
from sbi_amd import _lib

# SB_FAMILY=throughput switches the cooperative small-batch kernels (csrc/nsf_coop.h) off for an A/B run
if os.environ.get("SB_FAMILY") == "throughput":
    _lib.load().sbi_amd_nsf_set_coop_max_rows(0)
elif os.environ.get("SB_FAMILY") == "cooperative":
    _lib.load().sbi_amd_nsf_set_coop_max_rows(1 << 40)
print("family:", os.environ.get("SB_FAMILY", "default"), flush=True)
for B in batches:
    st = FusedTrainStep(est)
    idx = torch.randint(0, N, (B,), device="cuda")
    tb, xb = theta[idx].contiguous(), x[idx].contiguous()
    for _ in range(10):
        st.step(tb, xb)
    torch.cuda.synchronize()
    K = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K):
        st.step(tb, xb)
    e1.record()
    t_host = (time.perf_counter() - t0) / K * 1e3     # time to ENQUEUE
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / K * 1e3
    t_dev = e0.elapsed_time(e1) / K
    line = f"batch {B:6d}: enqueue {t_host:.3f} ms/step, wall {t_wall:.3f}, device span {t_dev:.3f}"
    if os.environ.get("SB_NO_GRAPH"):     # (rocprofv3 --kernel-trace hangs on the capture)
        print(line, flush=True)
        continue
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            st.step(tb, xb)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            st.step(tb, xb)
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            g.replay()
        torch.cuda.synchronize()
        line += f", graph replay {(time.perf_counter() - t0) / K * 1e3:.3f}"
    except Exception as e:  # noqa: BLE001
        line += f", graph capture failed: {type(e).__name__}: {str(e)[:200]}"
    print(line, flush=True)
