"""Why does bench.py's train leg report a slower fused step than tools/diag/small_batch.py?  Same estimator, same data,
200 steps each way; device span between two events around the whole loop."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from bench import make_data, build_estimator, TrainLeg
from sbi_amd.inference.trainers.fused import FusedTrainStep

dev = torch.device("cuda:0")
B = 65536
est = build_estimator(*make_data(B, "cpu"), dev)
th_all, x_all = make_data(90000, dev, seed=1000)

def span(fn, K=200, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(K): fn()
    e1.record(); host = (time.perf_counter() - t0) / K * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K, host

leg = TrainLeg(est, th_all, x_all, B, False, B)
d, h = span(leg); print(f"TrainLeg (gather + per-step events + step): {d:.4f} ms/step device span, host enqueue {h:.3f}; fused_ms per step {leg.fused_ms(200)/200:.4f}")
st = leg.stepper
tb, xb = th_all[:B].contiguous(), x_all[:B].contiguous()
d, h = span(lambda: st.step(tb, xb, global_batch=B)); print(f"same stepper, fixed batch, no events: {d:.4f} ms/step, host {h:.3f}")
def with_gather():
    th, xx = leg.sampler.batch(leg.calls, 0, B); leg.calls += 1
    st.step(th, xx, global_batch=B)
d, h = span(with_gather); print(f"gather + step, no events: {d:.4f} ms/step, host {h:.3f}")
def with_events():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); st.step(tb, xb, global_batch=B); e1.record()
d, h = span(with_events); print(f"fixed batch + per-step events: {d:.4f} ms/step, host {h:.3f}")
d, h = span(lambda: st.step(tb, xb, global_batch=B), K=20, warm=5); print(f"fixed batch, 20 steps: {d:.4f} ms/step")
