"""Child process of tests/test_dp_two_rank_gpu.py: the FUSED data-parallel training paths with world size 2 on the ONE
GPU a test box has.

    python tests/_dp_two_rank.py single <out.pt>            # no process group: the reference run
    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/_dp_two_rank.py dp <out_prefix>

Both ranks of the `dp` run sit on cuda:0 and talk through a `gloo` group (RCCL refuses two ranks on one device); the
gradient all-reduce of `FusedTrainStep` / `FusedFMPEStep` and the trainers' loss all-reduce are staged through the host
by sbi_amd/utils/collectives.py -- everything else is the product path: `rank_window` + `ShuffledGather.batch` +
`FusedTrainStep.step` with the `1 / global_batch` scaling inside the kernel (npe.py `launch_epoch`, fused.py
`loss_and_grad`), the rank-0 broadcasts of split / seeds / initial parameters, the pipelined epoch loop's snapshots.
The reference loop these replace: sbi/inference/trainers/base.py:1150-1193 (one device, no collective)."""
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
from sbi_amd.utils.shuffle import ShuffledGather, rank_window                       # noqa: E402
from tests.helpers import linear_gaussian_data                                      # noqa: E402

DEV = "cuda:0"


def _rw():
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def fused_nsf(global_batch, n_sims=6000, steps=5):
    """Steps of FusedTrainStep on this rank's window of the same global batches (ragged: 777 = 389 + 388 rows), the
    batches drawn by the device sampler exactly as NPE.train() draws them."""
    rank, world = _rw()
    theta, x = linear_gaussian_data(n_sims, 4, 3)
    torch.manual_seed(1)
    est = build_nsf(theta, x, hidden_features=32, num_transforms=3, num_bins=8).to(DEV)
    st = FusedTrainStep(est, lr=1e-3, clip_max_norm=5.0, distributed=world > 1)
    sampler = ShuffledGather(theta.to(DEV), x.to(DEV), None, seed=77)
    losses, rows = [], []
    for i in range(steps):
        lo, cnt = rank_window(i * global_batch, global_batch, rank, world)
        th, xx = sampler.batch(0, lo, cnt)
        losses.append(st.step(th, xx, global_batch=global_batch))
        rows.append(torch.arange(lo, lo + cnt))
    return {"params": est.net.flat_params.data.cpu(), "m": st.exp_avg.cpu(), "v": st.exp_avg_sq.cpu(),
            "losses": torch.cat(losses).cpu(), "rows": torch.cat(rows), "steps": steps, "grad": st.grad.cpu(),
            "grad_norm": st.grad_norm().cpu()}


def fused_fmpe():
    rank, world = _rw()
    theta, x = linear_gaussian_data(4000, 5, 4)
    torch.manual_seed(1)
    fm = build_flow_matching_estimator(theta, x).to(DEV)
    st = FusedFMPEStep(fm, lr=1e-3, clip_max_norm=5.0, distributed=world > 1)
    th, xx = theta.to(DEV), x.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    GB = 1001
    losses = []
    for i in range(3):
        t = torch.rand(GB, device=DEV, generator=g)           # the global batch's draws: every rank the same stream
        e = torch.randn(GB, 5, device=DEV, generator=g)
        lo, cnt = rank_window(0, GB, rank, world)
        s = slice(lo, lo + cnt)
        losses.append(st.loss_and_grad(th[i * GB:(i + 1) * GB][s].contiguous(), xx[i * GB:(i + 1) * GB][s].contiguous(),
                                       global_batch=GB, times=t[s].contiguous(), noise=e[s].contiguous()))
        st.apply()
    return {"params": fm.net.flat_params.data.cpu(), "m": st.exp_avg.cpu(), "v": st.exp_avg_sq.cpu(),
            "losses": torch.cat(losses).cpu()}


def npe_train():
    theta, x = linear_gaussian_data(4000, 3, 3)
    torch.manual_seed(2)
    inf = NPE(density_estimator="nsf", device=DEV, show_progress_bars=False)
    inf.append_simulations(theta, x)
    torch.manual_seed(5)      # split, sampler seed, permutation seed (rank 0's draws are broadcast)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.train(training_batch_size=501, max_num_epochs=6, stop_after_epochs=3)
    return {"params": est.net.flat_params.data.cpu(), "m": inf._stepper.exp_avg.cpu(), "v": inf._stepper.exp_avg_sq.cpu(),
            "train": torch.tensor(inf.summary["training_loss"], dtype=torch.float64),
            "val": torch.tensor(inf.summary["validation_loss"], dtype=torch.float64),
            "epochs": torch.tensor(inf.summary["epochs_trained"][-1]),
            "train_idx": inf.train_indices.cpu()}


def npe_round_two():
    """Multi-round NPE-C on the fused atomic step: each rank draws the contrasting atoms inside ITS share of a batch, so
    the run is not comparable to the one-rank run row for row; what must hold is that the replicas stay identical."""
    from torch.distributions import MultivariateNormal

    prior = MultivariateNormal(torch.zeros(3, device=DEV), 0.1 * torch.eye(3, device=DEV))
    proposal = MultivariateNormal(0.05 * torch.ones(3, device=DEV), 0.08 * torch.eye(3, device=DEV))
    theta, x = linear_gaussian_data(2000, 3, 3)
    torch.manual_seed(2)
    inf = NPE(prior=prior, density_estimator="nsf", device=DEV, show_progress_bars=False)
    torch.manual_seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=400, max_num_epochs=2)
        inf.append_simulations(theta[:1200] * 0.9, x[:1200], proposal=proposal)
        est = inf.train(training_batch_size=300, max_num_epochs=3, stop_after_epochs=5, num_atoms=6)
    return {"params": est.net.flat_params.data.cpu(),
            "val": torch.tensor(inf.summary["validation_loss"], dtype=torch.float64)}


def fmpe_train():
    theta, x = linear_gaussian_data(3000, 3, 3)
    torch.manual_seed(2)
    inf = FMPE(prior=None, device=DEV, show_progress_bars=False)
    inf.append_simulations(theta, x)
    torch.manual_seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.train(training_batch_size=400, max_num_epochs=4, stop_after_epochs=10**6, validation_times=3)
    return {"params": est.net.flat_params.data.cpu(),
            "train": torch.tensor(inf.summary["training_loss"], dtype=torch.float64),
            "val": torch.tensor(inf.summary["validation_loss"], dtype=torch.float64)}


def run_all():
    return {"fused_nsf_777": fused_nsf(777), "fused_nsf_1024": fused_nsf(1024),
            # 10 001 rows per rank: past the cooperative kernels' 8 192-row training limit -> the throughput kernels
            "fused_nsf_20002": fused_nsf(20002, n_sims=41000, steps=2), "fused_fmpe": fused_fmpe(),
            "npe_train": npe_train(), "npe_round_two": npe_round_two(), "fmpe_train": fmpe_train()}


def main():
    mode, out = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    if mode == "single":
        torch.save(run_all(), out)
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
        res = run_all()
    finally:
        dist.all_reduce, dist.broadcast = real_ar, real_bc
    res["_meta"] = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rank": rank, "calls": calls}
    torch.save(res, f"{out}.rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
