"""Repeat the fused loss/gradient pass on NaN-poisoned workspaces and report run-to-run differences."""
import sys, torch
sys.path.insert(0, "/root/repo")
from tests.helpers import matched_pair
from sbi_amd.inference.trainers.fused import FusedTrainStep
bad = 0
for cfg in [dict(D=10, C=10), dict(D=2, C=2), dict(D=4, C=7), dict(D=1, C=3), dict(D=5, C=4, num_bins=16, num_transforms=2)]:
    oracle, est, th, x = matched_pair(**cfg)
    for n in (777, 100, 65, 4096):
        theta, xx = th[:n].cuda() if n <= th.shape[0] else th.repeat(5, 1)[:n].cuda(), (x[:n].cuda() if n <= x.shape[0] else x.repeat(5, 1)[:n].cuda())
        st = FusedTrainStep(est)
        ref = None
        for it in range(12):
            st._workspace(n).fill_(float("nan"))
            l = st.loss_and_grad(theta, xx)
            torch.cuda.synchronize()
            g = st.grad.clone()
            if not torch.isfinite(g).all() or not torch.isfinite(l).all():
                print("NONFINITE", cfg, n, it); bad += 1
            if ref is None: ref = g
            elif not torch.equal(ref, g):
                print("NONDETERMINISTIC", cfg, n, it, (ref - g).abs().max().item()); bad += 1
print("bad", bad)
