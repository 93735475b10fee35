"""The device sampler's keyed permutation (csrc/shuffle.hip), restated in Python: a bijection of [0, n) for every n
and key, different per epoch key, and statistically an unremarkable shuffle (no CPU build of the kernel exists: the
kernel itself is held to this restatement index for index by tests/test_shuffle_gpu.py)."""

import numpy as np
import pytest

from sbi_amd.utils.shuffle import epoch_key, rank_window
from tests.shuffle_restatement import half_bits, prp


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 16, 17, 200, 1000, 4097])
def test_bijection(n):
    for key in (0, 1, 0xDEADBEEFCAFEF00D, epoch_key(123, 0), epoch_key(123, 1)):
        out = sorted(prp(i, n, key) for i in range(n))
        assert out == list(range(n)), (n, key)


def test_domain_is_less_than_four_times_n():
    for n in (1, 2, 5, 17, 90000, 100000, 2**20 + 1, 2**31 - 1):
        assert n <= 1 << (2 * half_bits(n)) <= max(4 * n, 16)


def test_epoch_keys_differ_and_orders_look_random():
    n = 2000
    keys = [epoch_key(7, e) for e in range(6)]
    assert len(set(keys)) == 6
    orders = np.array([[prp(i, n, k) for i in range(n)] for k in keys])
    for a in range(6):
        for b in range(a + 1, 6):
            assert (orders[a] == orders[b]).mean() < 0.01            # ~1/n fixed coincidences
    # position i is uncorrelated with where it lands; first-batch membership is spread over the range
    for o in orders:
        r = np.corrcoef(np.arange(n), o)[0, 1]
        assert abs(r) < 0.08
        first = np.sort(o[:200])
        assert first[0] < 100 and first[-1] > n - 100
    # over many keys every element lands in the first 10 % of the order about 10 % of the time
    hits = np.zeros(300)
    for e in range(400):
        k = epoch_key(99, e)
        head = [prp(i, 300, k) for i in range(30)]
        hits[head] += 1
    assert abs(hits.mean() - 40.0) < 1e-9 and hits.std() < 3.0 * np.sqrt(40 * 0.9)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("count", [1, 7, 200, 65536])
def test_rank_windows_tile_a_batch_like_the_index_split(world, count):
    """Data-parallel runs: rank r takes a contiguous window of every global batch -- the windows tile the batch exactly, in
    rank order, and equal the slice the index path (`my_slice`: ceil(count / world) rows per rank) takes."""
    lo = 1000
    idx = list(range(lo, lo + count))
    per = (count + world - 1) // world
    covered = []
    for r in range(world):
        off, rows = rank_window(lo, count, r, world)
        assert rows >= 0 and (rows == 0 or lo <= off < lo + count)
        want = idx[r * per : min((r + 1) * per, count)]
        assert list(range(off, off + rows)) == want
        covered += want
    assert covered == idx
