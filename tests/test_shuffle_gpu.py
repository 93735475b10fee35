"""The device-side minibatch sampler (csrc/shuffle.hip behind sbi_amd_shuffled_gather) against its Python restatement,
index for index, and as a sampler: every epoch's batches tile the training split exactly once."""

import pytest
import torch

from sbi_amd.utils.shuffle import ShuffledGather, epoch_key
from tests.shuffle_restatement import prp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 5, 17, 200, 1000, 4097])
def test_kernel_matches_restatement(n):
    th = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3).cuda()
    xx = (torch.arange(n * 2, dtype=torch.float32).reshape(n, 2) * 0.5).cuda()
    sg = ShuffledGather(th, xx, None, seed=11)
    for epoch in (0, 1, 5):
        idx = sg.indices(epoch, 0, n).cpu().tolist()
        assert idx == [prp(i, n, epoch_key(11, epoch)) for i in range(n)]
        a, b = sg.batch(epoch, 0, n)
        assert torch.equal(a.cpu(), th.cpu()[idx]) and torch.equal(b.cpu(), xx.cpu()[idx])
    # a window of the order, through a base index (the training split's row numbers)
    base = torch.randperm(n)
    sg2 = ShuffledGather(th, xx, base.cuda(), seed=3)
    lo, cnt = n // 3, n - n // 3
    idx2 = sg2.indices(2, lo, cnt).cpu()
    want = base[[prp(i, n, epoch_key(3, 2)) for i in range(lo, lo + cnt)]]
    assert torch.equal(idx2, want)
    a2, _ = sg2.batch(2, lo, cnt)
    assert torch.equal(a2.cpu(), th.cpu()[want])


def test_epoch_batches_tile_the_split_once_at_bench_size():
    n, B = 90000, 8192
    th = torch.randn(n, 10).cuda()
    xx = torch.randn(n, 10).cuda()
    sg = ShuffledGather(th, xx, None, seed=2026)
    for epoch in (0, 1):
        parts = [sg.indices(epoch, lo, min(B, n - lo)) for lo in range(0, n, B)]
        allidx = torch.cat(parts)
        assert torch.equal(torch.sort(allidx).values, torch.arange(n, device="cuda"))
    assert (sg.indices(0, 0, n) != sg.indices(1, 0, n)).float().mean().item() > 0.99
    a, b = sg.batch(0, 65536 - 100, 200)
    i = sg.indices(0, 65536 - 100, 200)
    assert torch.equal(a, th[i]) and torch.equal(b, xx[i])


def test_refusals():
    th = torch.randn(10, 3).cuda()
    xx = torch.randn(10, 2).cuda()
    sg = ShuffledGather(th, xx, None, seed=1)
    with pytest.raises(RuntimeError):
        sg.batch(0, 5, 6)            # window past the end of the order
    with pytest.raises(Exception):
        ShuffledGather(th.cpu(), xx.cpu(), None, seed=1)
    assert sg.batch(0, 10, 0)[0].shape == (0, 3)
