"""Differential check of the host helpers rewritten in round 5 (sbi_amd/utils/sbiutils.py: gradient_ascent, mcmc_transform,
z-score statistics, handle_invalid_x) against the REAL reference functions, imported from /root/reference with the absent
third-party packages stubbed (tools/make_golden.py).  Build container only.  Prints ALL OK."""
import os, sys, warnings, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, ROOT)
import make_golden  # stubs + reference on path
from sbi.utils import sbiutils as ref
from sbi_amd.utils import sbiutils as mine
from torch.distributions import MultivariateNormal, Independent, Uniform, Gamma, Normal
warnings.simplefilter("ignore")
ok=True
# gradient_ascent with transforms (bounded + unbounded), various settings
for seed in range(4):
    for prior in (MultivariateNormal(torch.tensor([0.5,-1.0]), torch.diag(torch.tensor([2.0,0.5]))), Independent(Uniform(-2*torch.ones(2), 3*torch.ones(2)),1)):
        torch.manual_seed(seed)
        tf_r = ref.mcmc_transform(prior); torch.manual_seed(seed); tf_m = mine.mcmc_transform(prior)
        th = prior.sample((7,))
        assert torch.allclose(tf_r(th), tf_m(th)), "transform differs"
        target = MultivariateNormal(torch.tensor([0.3,0.4]), torch.tensor([[0.3,0.1],[0.1,0.2]]))
        pot = lambda t: target.log_prob(t)
        inits = prior.sample((150,))
        for kw in (dict(num_iter=37, num_to_optimize=12, learning_rate=0.05, save_best_every=10), dict(num_iter=20, num_to_optimize=300, learning_rate=0.02, save_best_every=7)):
            a_r, v_r = ref.gradient_ascent(pot, inits.clone(), theta_transform=tf_r, **kw)
            a_m, v_m = mine.gradient_ascent(pot, inits.clone(), theta_transform=tf_m, **kw)
            same = torch.allclose(a_r.reshape(-1), a_m.reshape(-1), atol=1e-6) and torch.allclose(v_r.reshape(-1), v_m.reshape(-1), atol=1e-6)
            print(type(prior).__name__, seed, kw['num_iter'], "same" if same else f"DIFF {a_r} {a_m} {v_r} {v_m}")
            ok &= same
# z-score helpers
for structured in (False, True):
    x = torch.randn(50, 6) * 3 + 1; x[3,2] = float('nan'); x[7,0]=float('inf')
    for f in ("z_standardization",):
        r = getattr(ref,f)(x, structured); m = getattr(mine,f)(x, structured)
        assert all(torch.allclose(a,b, equal_nan=True) for a,b in zip(r,m)), f
    r = ref.handle_invalid_x(x, True); m = mine.handle_invalid_x(x, True)
    assert torch.equal(r[0], m[0]) and r[1:]==m[1:], (r[1:], m[1:])
print("within_support:", torch.equal(ref.within_support(Independent(Uniform(-torch.ones(2), torch.ones(2)),1), torch.randn(20,2)), mine.within_support(Independent(Uniform(-torch.ones(2), torch.ones(2)),1), torch.randn(20,2))) or "rng differs (expected)")
print("ALL OK" if ok else "MISMATCH")
