"""csrc/compact.hip through the C ABI and through `accept_reject_sample`: the device loop must return exactly what the
torch (host) loop returns -- same rows, same order, same acceptance rates -- for the mask and the fused-box acceptance,
one and several conditions, requests that end inside a batch, and batches larger than one scan tile
(reference: sbi/samplers/rejection/rejection.py:368-409)."""
import numpy as np
import pytest
import torch

from sbi_amd import _lib
from sbi_amd.samplers.rejection.rejection import accept_reject_sample

pytestmark = pytest.mark.gpu


def _reference(cand, acc, filled, num_samples):
    bs, X, ev = cand.shape
    out = np.full((num_samples, X, ev), np.nan, np.float32)
    f = filled.copy()
    for x in range(X):
        rows = cand[acc[:, x], x]
        take = rows[: max(0, num_samples - f[x])]
        out[f[x] : f[x] + len(take), x] = take
        f[x] = min(num_samples, f[x] + len(rows))
    return out, f, acc.sum(0)


@pytest.mark.parametrize("bs,X,ev,p,box", [(1, 1, 3, 0.5, False), (1000, 1, 10, 0.3, False), (5000, 3, 4, 0.9, False),
                                           (70000, 1, 10, 0.98, True), (20000, 2, 5, 0.05, True), (4097, 1, 1, 1.0, True),
                                           (4096, 1, 2, 0.0, False), (300000, 1, 10, 0.5, False),
                                           (2500, 1, 32, 0.7, True), (2500, 1, 33, 0.7, True), (100001, 1, 7, 0.31, True)])
def test_c_abi_matches_a_host_compaction(bs, X, ev, p, box):
    lib = _lib.load()
    g = torch.Generator().manual_seed(bs + X)
    cand = torch.rand(bs, X, ev, generator=g) * 2 - 1
    if box:
        width = p ** (1.0 / ev)
        lo, hi = torch.full((ev,), -width), torch.full((ev,), width)
        cand[::97, :, 0] = float("nan")                       # NaN rows fail the interval check
        acc = ((cand >= lo) & (cand <= hi)).all(-1)
    else:
        acc = torch.rand(bs, X, generator=g) < p
    num_samples = max(1, int(0.6 * bs * max(p, 0.01)))
    filled0 = np.array([2, 3, 1][:X], np.int64) if bs > 1000 else np.array([0, 3, 1][:X], np.int64)
    want, f_want, n_acc = _reference(cand.numpy(), acc.numpy(), filled0, num_samples)
    d = "cuda"
    out = torch.full((num_samples, X, ev), float("nan"), device=d)
    state = torch.zeros(3 * X, dtype=torch.long, device=d)
    state[:X] = torch.from_numpy(filled0)
    control = torch.zeros(2 * X, dtype=torch.int32, device=d)
    scan = torch.zeros(int(lib.sbi_amd_accept_compact_scan_words(bs, X)), dtype=torch.long, device=d)
    cc, mk = cand.to(d), acc.to(d)
    lo_d, hi_d = (lo.to(d), hi.to(d)) if box else (None, None)
    for gen in (1, 2):            # second call on the same scan buffer with the next generation: same answer
        state[:X] = torch.from_numpy(filled0).to(d)
        state[X:] = 0
        out.fill_(float("nan"))
        rc = lib.sbi_amd_accept_compact(_lib.ptr(cc), None if box else _lib.ptr(mk), _lib.ptr(lo_d), _lib.ptr(hi_d), bs, X, ev, _lib.ptr(out), num_samples,
                                        _lib.ptr(state), _lib.ptr(control), _lib.ptr(scan), gen,
                                        _lib.current_stream(torch.device(d)))
        assert rc == 0
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for x in range(X):
            lo_r, hi_r = filled0[x], f_want[x]
            assert np.array_equal(got[lo_r:hi_r, x], want[lo_r:hi_r, x]), (gen, x)
        st = state.cpu().numpy()
        assert np.array_equal(st[:X], f_want) and np.array_equal(st[X : 2 * X], n_acc) and np.array_equal(st[2 * X :], n_acc)
        assert int(control.abs().sum()) == 0


class _Replay:
    """A proposal that hands out pre-drawn batches (so the device and the host loop see identical candidates)."""

    def __init__(self, X, ev, device, seed):
        self.g, self.X, self.ev, self.device = torch.Generator().manual_seed(seed), X, ev, device

    def __call__(self, shape, condition=None):
        return (torch.randn(shape[0], self.X, self.ev, generator=self.g) * 0.8).to(self.device)


@pytest.mark.parametrize("X,ev,num,cap,fused", [(1, 10, 100_000, 1_000_000, True), (1, 3, 5000, 700, False),
                                               (4, 2, 3000, 1000, True), (2, 5, 2500, 10_000, False)])
def test_device_loop_equals_host_loop(X, ev, num, cap, fused):
    lo, hi = torch.full((ev,), -1.0), torch.full((ev,), 1.2)

    def accept_on(dev):
        def fn(c):
            return ((c >= lo.to(dev)) & (c <= hi.to(dev))).all(-1)

        if fused:
            fn.box_bounds = (lo.to(dev), hi.to(dev))
        return fn

    cond = torch.zeros(X, 1)
    s_dev, a_dev = accept_reject_sample(_Replay(X, ev, "cuda", 5), accept_on("cuda"), num, max_sampling_batch_size=cap,
                                        proposal_sampling_kwargs=dict(condition=cond))
    s_cpu, a_cpu = accept_reject_sample(_Replay(X, ev, "cpu", 5), accept_on("cpu"), num, max_sampling_batch_size=cap,
                                        proposal_sampling_kwargs=dict(condition=cond))
    assert s_dev.shape == s_cpu.shape == (num, X, ev)
    assert torch.equal(s_dev.cpu(), s_cpu) and torch.allclose(a_dev.cpu(), a_cpu)
