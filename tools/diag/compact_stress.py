"""Stress of sbi_amd_accept_compact against a torch compaction: random sizes, widths, acceptance rates, offsets, mask and
box acceptance, reusing one scan buffer across calls (generation counter).  usage: python tools/diag/compact_stress.py [n]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd import _lib

lib = _lib.load()
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
max_bs = 400_000
scan = torch.zeros(int(lib.sbi_amd_accept_compact_scan_words(max_bs, 1)), dtype=torch.long, device=d)
control = torch.zeros(2, dtype=torch.int32, device=d)
gen = 0
for case in range(n_cases):
    bs = int(torch.randint(1, max_bs, (1,), generator=g)) if case % 3 else int(torch.randint(1, 3000, (1,), generator=g))
    ev = int(torch.randint(1, 40, (1,), generator=g))
    p = float(torch.rand(1, generator=g))
    box = bool(case % 2)
    cand = (torch.rand(bs, 1, ev, generator=g) * 2 - 1).to(d)
    if box:
        w = max(p, 1e-3) ** (1.0 / ev)
        lo, hi = torch.full((ev,), -w, device=d), torch.full((ev,), w, device=d)
        acc = ((cand >= lo) & (cand <= hi)).all(-1)
        mk = None
    else:
        acc = (torch.rand(bs, 1, generator=g) < p).to(d)
        mk, lo, hi = acc.contiguous(), None, None
    filled0 = int(torch.randint(0, 50, (1,), generator=g))
    num_samples = max(1, int(torch.randint(1, bs + 60, (1,), generator=g)))
    filled0 = min(filled0, num_samples)
    out = torch.full((num_samples, 1, ev), float("nan"), device=d)
    state = torch.tensor([filled0, 0, 0], dtype=torch.long, device=d)
    gen += 1
    rc = lib.sbi_amd_accept_compact(_lib.ptr(cand), _lib.ptr(mk), _lib.ptr(lo), _lib.ptr(hi), bs, 1, ev, _lib.ptr(out),
                                    num_samples, _lib.ptr(state), _lib.ptr(control), _lib.ptr(scan), gen,
                                    _lib.current_stream(d))
    assert rc == 0, rc
    rows = cand[acc[:, 0], 0]
    take = rows[: max(0, num_samples - filled0)]
    want = torch.full((num_samples, 1, ev), float("nan"), device=d)
    want[filled0 : filled0 + len(take), 0] = take
    st = state.cpu().tolist()
    ok = torch.equal(torch.nan_to_num(out, nan=123.0), torch.nan_to_num(want, nan=123.0))
    assert ok and st == [min(num_samples, filled0 + len(rows)), len(rows), len(rows)] and control.cpu().tolist() == [0, 0], \
        (case, bs, ev, p, box, st, filled0, len(rows), num_samples)
print(f"{n_cases} cases ok")
