"""Child process of tests/test_rccl_one_rank_gpu.py (and of `bench.py`'s `rccl_1rank` leg is NOT this file).

Runs every fused training path twice on the ONE visible GPU -- first without a process group, then as rank 0 of a
1-rank `nccl` (= RCCL) group, with identical seeds -- and writes what it measured as JSON to argv[1].  With one rank the
all-reduce of the flat gradient is the identity, so the two runs must agree BIT FOR BIT: parameters, Adam moments,
per-row losses, loss history.  What this executes on real hardware: RCCL communicator init on the device, the
rank-0 broadcasts of `NPE.train()` (split indices, permutation seed, initial parameters), the `all_reduce` on the
flat device gradient buffer between the fused backward pass and the fused clip + Adam kernel, the per-epoch loss
all-reduce, `snapshot / restore_optimizer` of the pipelined epoch loop under DP."""
import json
import os
import sys
import warnings

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sbi_amd.inference import FMPE, NPE                                             # noqa: E402
from sbi_amd.inference.trainers.fused import FusedFMPEStep, FusedTrainStep          # noqa: E402
from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator   # noqa: E402
from sbi_amd.neural_nets.net_builders.flow import build_nsf                         # noqa: E402
from tests.helpers import linear_gaussian_data                                      # noqa: E402

DEV = "cuda:0"


def fused_nsf(distributed):
    theta, x = linear_gaussian_data(3000, 4, 3)
    torch.manual_seed(1)
    est = build_nsf(theta, x, hidden_features=32, num_transforms=3, num_bins=8).to(DEV)
    st = FusedTrainStep(est, lr=1e-3, clip_max_norm=5.0, distributed=distributed)
    th, xx = theta.to(DEV), x.to(DEV)
    losses = []
    for i in range(4):
        s = slice(500 * i, 500 * i + 777)           # ragged batch
        losses.append(st.step(th[s].contiguous(), xx[s].contiguous(), global_batch=777))
    snap = st.snapshot()
    st.step(th[:512].contiguous(), xx[:512].contiguous())
    st.restore_optimizer(snap)
    return {"params": est.net.flat_params.data.cpu(), "m": st.exp_avg.cpu(), "v": st.exp_avg_sq.cpu(),
            "losses": torch.cat(losses).cpu(), "step": st.step_count, "grad_norm": st.grad_norm().cpu()}


def fused_fmpe(distributed):
    theta, x = linear_gaussian_data(2000, 5, 4)
    torch.manual_seed(1)
    fm = build_flow_matching_estimator(theta, x).to(DEV)
    st = FusedFMPEStep(fm, lr=1e-3, clip_max_norm=5.0, distributed=distributed)
    th, xx = theta.to(DEV), x.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    losses = []
    for i in range(3):
        t = torch.rand(1000, device=DEV, generator=g)
        e = torch.randn(1000, 5, device=DEV, generator=g)
        losses.append(st.loss_and_grad(th[:1000], xx[:1000], times=t, noise=e))
        st.apply()
    return {"params": fm.net.flat_params.data.cpu(), "m": st.exp_avg.cpu(), "v": st.exp_avg_sq.cpu(),
            "losses": torch.cat(losses).cpu()}


def npe_train():
    theta, x = linear_gaussian_data(3000, 3, 3)
    torch.manual_seed(2)
    inf = NPE(density_estimator="nsf", device=DEV, show_progress_bars=False)
    inf.append_simulations(theta, x)
    torch.manual_seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.train(training_batch_size=500, max_num_epochs=6, stop_after_epochs=2)
    return {"params": est.net.flat_params.data.cpu(), "m": inf._stepper.exp_avg.cpu(),
            "v": inf._stepper.exp_avg_sq.cpu(), "train": torch.tensor(inf.summary["training_loss"]),
            "val": torch.tensor(inf.summary["validation_loss"]), "epochs": inf.summary["epochs_trained"][-1]}


def fmpe_train():
    theta, x = linear_gaussian_data(2000, 3, 3)
    torch.manual_seed(2)
    inf = FMPE(prior=None, device=DEV, show_progress_bars=False)
    inf.append_simulations(theta, x)
    torch.manual_seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.train(training_batch_size=400, max_num_epochs=4, stop_after_epochs=10**6, validation_times=3)
    return {"params": est.net.flat_params.data.cpu(), "train": torch.tensor(inf.summary["training_loss"]),
            "val": torch.tensor(inf.summary["validation_loss"])}


def compare(a, b):
    out = {}
    for k in a:
        va, vb = a[k], b[k]
        if isinstance(va, torch.Tensor):
            same = va.shape == vb.shape and bool(torch.equal(va, vb))
            out[k] = {"bit_identical": same,
                      "max_abs_diff": float((va.double() - vb.double()).abs().max()) if va.shape == vb.shape else None,
                      "finite": bool(torch.isfinite(va).all())}
        else:
            out[k] = {"bit_identical": va == vb, "a": va, "b": vb}
    return out


def main():
    torch.cuda.set_device(0)
    single = {"fused_nsf": fused_nsf(False), "fused_fmpe": fused_fmpe(False), "npe_train": npe_train(),
              "fmpe_train": fmpe_train()}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    probe = torch.ones(3, device=DEV)
    dist.all_reduce(probe)
    calls = {"all_reduce": 0, "broadcast": 0}
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    def ar(*a, **k):
        calls["all_reduce"] += 1
        return real_ar(*a, **k)

    def bc(*a, **k):
        calls["broadcast"] += 1
        return real_bc(*a, **k)

    dist.all_reduce, dist.broadcast = ar, bc
    try:
        rccl = {"fused_nsf": fused_nsf(True), "fused_fmpe": fused_fmpe(True), "npe_train": npe_train(),
                "fmpe_train": fmpe_train()}
    finally:
        dist.all_reduce, dist.broadcast = real_ar, real_bc
    report = {"backend": dist.get_backend(), "world": dist.get_world_size(), "probe_sum": float(probe[0]),
              "collective_calls": calls, "legs": {k: compare(single[k], rccl[k]) for k in single}}
    dist.destroy_process_group()
    with open(sys.argv[1], "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
