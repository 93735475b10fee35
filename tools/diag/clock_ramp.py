"""How the device's speed depends on what it did just before: per-step device time of the fused 65 536-row training
step over a run with idle gaps of different lengths in it.  usage: python tools/diag/clock_ramp.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import make_data, build_estimator
from sbi_amd.inference.trainers.fused import FusedTrainStep

dev = torch.device("cuda:0")
theta, x = make_data(65536, dev)
est = build_estimator(*make_data(65536, "cpu"), dev)
st = FusedTrainStep(est)


def burst(n):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        st.step(theta, x)
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]


def show(tag, t):
    k = len(t)
    groups = [t[0:5], t[5:10], t[10:20], t[20:40], t[40:80], t[80:160], t[160:]]
    print(f"{tag:34s}" + " | ".join(f"{sum(g) / len(g):.3f}" for g in groups if g), flush=True)


print("mean ms per step over steps [0,5) [5,10) [10,20) [20,40) [40,80) [80,160) [160,..)")
show("cold start, 300 steps", burst(300))
for gap in (0.0, 0.002, 0.01, 0.03, 0.1, 0.5):
    time.sleep(gap)
    show(f"after {gap * 1e3:.0f} ms idle, 100 steps", burst(100))
# the bench's sequence: warm-up, synchronize, a host pause of the length of gc.collect(), 20 timed steps
import gc
burst(60)
t0 = time.perf_counter(); gc.collect(); g = time.perf_counter() - t0
show(f"after gc.collect() ({g * 1e3:.0f} ms), 20 steps", burst(20))
burst(60)
show("directly after 60 steps, 20 steps", burst(20))
