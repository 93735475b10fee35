"""Data-parallel path on CPU: world_size 2, gloo.  Each rank takes its slice of every
global batch, gradients are summed with all-reduce, and both ranks must end with
identical parameters that match single-process training on the full batches."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sbi_amd.inference import NPE
from tests.helpers import linear_gaussian_data
from tests.oracle_adapter import oracle_build_fn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train(seed_model=2, epochs=3):
    theta, x = linear_gaussian_data(400, 2, 2)
    torch.manual_seed(seed_model)
    inf = NPE(density_estimator=oracle_build_fn(hidden_features=16, num_transforms=2, num_bins=4),
              show_progress_bars=False)
    inf.append_simulations(theta, x)
    torch.manual_seed(5)     # split / epoch permutations (rank 0's stream is broadcast)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.train(training_batch_size=90, max_num_epochs=epochs)
    return torch.cat([p.detach().reshape(-1) for p in est.parameters()]), inf.summary


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    flat, summary = _train()
    out[rank] = (flat, summary["validation_loss"])
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_data_parallel_matches_single_process():
    torch.set_num_threads(1)
    ref_flat, ref_summary = _train()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    f0, v0 = out[0]
    f1, v1 = out[1]
    assert torch.equal(f0, f1), "replicas diverged"
    assert v0 == v1
    # same global batches, gradient = sum of the two half-batch gradients: equal up to fp32 re-association
    assert (f0 - ref_flat).abs().max() < 2e-4
    assert abs(v0[-1] - ref_summary["validation_loss"][-1]) < 1e-3


# ---- the FMPE trainer's data-parallel plumbing (same contract: slices, summed gradients, identical replicas)
def _train_fmpe(epochs=3):
    from sbi_amd.inference import FMPE
    from tests.oracle_adapter import oracle_vf_build_fn

    theta, x = linear_gaussian_data(400, 3, 2)
    torch.manual_seed(2)
    inf = FMPE(vf_estimator=oracle_vf_build_fn(H=16, L=2, E=8), show_progress_bars=False)
    inf.append_simulations(theta, x)
    torch.manual_seed(5)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.train(training_batch_size=90, max_num_epochs=epochs, validation_times=3)
    return torch.cat([p.detach().reshape(-1) for p in est.parameters()]), inf.summary


def _worker_fmpe(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    flat, summary = _train_fmpe()
    out[rank] = (flat, summary["validation_loss"])
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_fmpe_two_rank_data_parallel_matches_single_process():
    torch.set_num_threads(1)
    ref_flat, ref_summary = _train_fmpe()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_fmpe, args=(2, _free_port(), out), nprocs=2, join=True)
    f0, v0 = out[0]
    f1, v1 = out[1]
    assert torch.equal(f0, f1), "replicas diverged"
    assert (f0 - ref_flat).abs().max() <= 1e-5 * max(1.0, ref_flat.abs().max().item())
    assert len(v0) == len(ref_summary["validation_loss"])
    assert max(abs(a - b) for a, b in zip(v0, ref_summary["validation_loss"])) < 1e-5


def test_collectives_choose_the_staged_path_by_capability_not_by_name():
    """a group whose backend string is not literally "nccl" (the composite `cpu:gloo,cuda:nccl` of a bare
    `init_process_group()`, or something unknown) must get device tensors as they are; only pure gloo is staged through
    the host, and says so once."""
    import warnings

    from sbi_amd.utils import collectives

    class FakeDist:
        def __init__(self, backend):
            self.backend = backend

        def get_backend(self, group=None):
            return self.backend

    class FakeDeviceTensor:       # (no ROCm device in the CPU suite: only `.device.type` is looked at)
        class device:
            type = "cuda"

    t = FakeDeviceTensor()
    for backend in ("nccl", "cpu:gloo,cuda:nccl", "NCCL", "ucc", "custom"):
        assert collectives._direct(FakeDist(backend), t), backend
    collectives._warned_staged = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert not collectives._direct(FakeDist("gloo"), t)
        assert not collectives._direct(FakeDist("gloo"), t)
    assert sum("staged through the host" in str(x.message) for x in w) == 1
    import torch

    assert collectives._direct(FakeDist("gloo"), torch.zeros(1))      # host tensors never need staging
    # a per-device map is judged by its cuda entry: gloo or nothing for cuda means staging
    for backend in ("cpu:gloo,cuda:gloo", "cpu:gloo"):
        assert not collectives.device_capable(FakeDist(backend)), backend
    assert collectives.device_capable(FakeDist("cpu:gloo,cuda:nccl"))
