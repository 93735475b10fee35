"""Child process of tests/test_native_allreduce_gpu.py: rank 0 of a ONE-rank group on the box's single GPU drives the
fused NSF training step with the gradient all-reduce going through `sbi_amd_allreduce_flat` (RCCL resolved by the
kernel library) and compares with the same steps (i) without any process group and (ii) through
torch.distributed's nccl backend.  Writes JSON to argv[1]."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sbi_amd.inference.trainers.fused import FusedTrainStep          # noqa: E402
from sbi_amd.neural_nets.net_builders.flow import build_nsf          # noqa: E402
from sbi_amd.utils.collectives import NativeAllReduce                # noqa: E402
from tests.helpers import linear_gaussian_data                       # noqa: E402

DEV = "cuda:0"


def run(distributed, native):
    theta, x = linear_gaussian_data(3000, 10, 10)
    torch.manual_seed(1)
    est = build_nsf(theta, x).to(DEV)
    st = FusedTrainStep(est, lr=1e-3, clip_max_norm=5.0, distributed=distributed, native_allreduce=native)
    th, xx = theta.to(DEV), x.to(DEV)
    losses = [st.step(th[500 * i : 500 * i + 777].contiguous(), xx[500 * i : 500 * i + 777].contiguous(), global_batch=777)
              for i in range(4)]
    return {"params": est.net.flat_params.data.cpu(), "m": st.exp_avg.cpu(), "losses": torch.cat(losses).cpu()}


def main(out_path):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    plain = run(False, False)
    dist.init_process_group("gloo", rank=0, world_size=1)       # (only ships the communicator id)
    native = run(True, True)
    rep = {"bit_identical": {k: bool(torch.equal(plain[k], native[k])) for k in plain}}
    # the bare collective: value and latency on the 98 025-float gradient of the default network
    ar = NativeAllReduce(dist, DEV)
    buf = torch.randn(98025, device=DEV)
    ref = buf.clone()
    ar(buf)
    torch.cuda.synchronize()
    rep["one_rank_identity"] = bool(torch.equal(buf, ref))
    for _ in range(20):
        ar(buf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ar(buf)
    torch.cuda.synchronize()
    rep["allreduce_us_98025_floats"] = (time.perf_counter() - t0) / 200 * 1e6
    ar.close()
    dist.destroy_process_group()
    json.dump(rep, open(out_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1])
