#!/usr/bin/env python
"""tests/golden/rejection_reference.pt: outputs of the reference's REAL `rejection_sample` and `gradient_ascent`
(sbi/samplers/rejection/rejection.py:18-227, sbi/utils/sbiutils.py:1160-1286; pure torch, importable once the absent
third-party packages are stubbed) on analytic potentials, seeded.  tests/test_golden_rejection.py replays them against
sbi_amd's implementations with the same seeds (same RNG call order => same candidates, same accept decisions).
Run in the build container only (needs /root/reference)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden  # noqa: E402,F401  (installs the stubs, puts /root/reference on sys.path)

from torch.distributions import Independent, MultivariateNormal, Normal, Uniform  # noqa: E402

from sbi.samplers.rejection.rejection import rejection_sample  # noqa: E402
from sbi.utils.sbiutils import gradient_ascent  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "rejection_reference.pt")


def cases():
    # (name, potential, proposal, kwargs)
    target = MultivariateNormal(torch.tensor([0.3, -0.2, 0.1]), 0.05 * torch.eye(3))
    prop = MultivariateNormal(torch.zeros(3), 0.3 * torch.eye(3))
    yield "gauss_in_gauss", (lambda th: target.log_prob(th) + 1.7), prop, dict(
        num_samples=700, max_sampling_batch_size=300, num_samples_to_find_max=500, num_iter_to_find_max=30, m=1.2)
    box = Independent(Uniform(-1.5 * torch.ones(2), 1.5 * torch.ones(2)), 1)
    t2 = MultivariateNormal(torch.tensor([0.5, -0.4]), torch.tensor([[0.08, 0.03], [0.03, 0.05]]))
    yield "gauss_in_box", (lambda th: t2.log_prob(th)), box, dict(
        num_samples=400, max_sampling_batch_size=10_000, num_samples_to_find_max=300, num_iter_to_find_max=20, m=1.5)


def main():
    g = {}
    for name, pot, prop, kw in cases():
        torch.manual_seed(7)
        samples, acc = rejection_sample(pot, prop, **kw)
        g[name] = dict(kw=kw, samples=samples.clone(), acceptance=torch.as_tensor(acc).clone())
    # gradient_ascent alone: value and argmax of a smooth potential
    torch.manual_seed(3)
    t = MultivariateNormal(torch.tensor([1.0, -2.0]), torch.tensor([[0.5, 0.1], [0.1, 0.3]]))
    inits = torch.randn(200, 2) * 2
    arg, val = gradient_ascent(lambda th: t.log_prob(th), inits, num_iter=60, num_to_optimize=20, learning_rate=0.05)
    g["gradient_ascent"] = dict(inits=inits, argmax=arg.detach().clone(), max=val.detach().clone())
    torch.save(g, OUT)
    print("wrote", OUT, {k: (v["samples"].shape if "samples" in v else None) for k, v in g.items()})


if __name__ == "__main__":
    main()
