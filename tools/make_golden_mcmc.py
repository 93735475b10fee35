#!/usr/bin/env python
"""Generate tests/golden/mcmc_reference.pt from the REAL reference code (build container only).

Runs sbi's own `SliceSamplerVectorized` (sbi/samplers/mcmc/slice_numpy.py:353-587, pure numpy, in the
reference tree) one chain at a time with its `rng` replaced by a replayer that serves a recorded table of
uniforms in the layout `sbi_amd_mcmc_slice_tick` consumes (per tick and chain: u0 -> log u, u1 -> bracket
offset, u2 -> point in the bracket, u[4+d] -> Fisher-Yates dimension shuffle), and `mcmc_transform`
(sbi/utils/sbiutils.py:867-980) on a Gaussian and a box prior.  tests/test_golden_mcmc.py replays the same
table through the HIP tick kernel; /root/reference is NOT needed at test time.
"""

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden  # noqa: E402,F401  (installs the third-party stubs and puts /root/reference on sys.path)

for mod in ["matplotlib", "matplotlib.pyplot", "matplotlib.axes", "matplotlib.figure", "joblib"]:
    try:
        __import__(mod)
    except Exception:
        make_golden.stub(mod)

OUT = make_golden.OUT
WEIGHTS = torch.tensor([1.0, 0.25, 4.0, 0.5])
CENTER = 0.3


def log_prob_np(params: np.ndarray) -> np.ndarray:
    """float32 torch arithmetic, as the potential is evaluated in the product"""
    th = torch.as_tensor(params, dtype=torch.float32)
    return (-0.5 * ((th - CENTER) ** 2 / WEIGHTS).sum(1)).numpy()


class ReplayRNG:
    """Stands in for np.random inside ONE single-chain reference sampler."""

    def __init__(self, table: np.ndarray, sampler_ref):
        self.table, self.sampler_ref, self.tick, self.calls = table, sampler_ref, -1, 0

    def new_tick(self):
        self.tick += 1
        self.calls = 0

    def rand(self):
        u = self.table[self.tick]
        state = self.sampler_ref[0].state[0]["state"]
        if state == "BEGIN":
            v = u[0] if self.calls == 0 else u[1]
        else:
            v = u[2]
        self.calls += 1
        return float(v)

    def shuffle(self, order):
        if self.tick < 0:          # the initial order is given to both implementations explicitly
            return
        u = self.table[self.tick]
        D = len(order)
        order[:] = list(range(D))
        for d in range(D - 1, 0, -1):
            k = min(int(np.float32(u[4 + d]) * np.float32(d + 1)), d)
            order[d], order[k] = order[k], order[d]


def main():
    from sbi.samplers.mcmc.slice_numpy import SliceSamplerVectorized
    from sbi.utils.sbiutils import mcmc_transform
    from sbi.utils.torchutils import BoxUniform

    torch.manual_seed(11)
    C, D, NS, TUNE, TICKS = 24, 4, 5, 3, 3000
    init = torch.randn(C, D)
    order0 = torch.rand(C, D).argsort(1)
    table = torch.rand(TICKS, C, 4 + D)
    samples, widths, ticks_used = [], [], []
    for c in range(C):
        holder = []
        s = SliceSamplerVectorized(log_prob_fn=None, init_params=init[c : c + 1].numpy().astype(np.float64),
                                   num_chains=1, thin=1, tuning=TUNE, verbose=False, init_width=0.7)
        holder.append(s)
        rng = ReplayRNG(table[:, c].numpy(), holder)
        s.rng = rng

        def lp(params, rng=rng):
            rng.new_tick()
            return log_prob_np(params)

        s._log_prob_fn = lp
        # run() shuffles the initial order through self.rng (a no-op here) -> install ours afterwards is not
        # possible, so patch list(range) result by pre-seeding: run() builds order = list(range(D)) then shuffles
        orig_shuffle = rng.shuffle

        def first_shuffle(order, c=c):
            order[:] = order0[c].tolist()
            rng.shuffle = orig_shuffle

        rng.shuffle = first_shuffle
        out = s.run(NS)                       # (1, NS, D)
        samples.append(torch.as_tensor(out[0], dtype=torch.float64))
        widths.append(torch.as_tensor(s.state[0]["width"], dtype=torch.float64))
        ticks_used.append(rng.tick + 1)
    g = dict(C=C, D=D, num_samples=NS, tuning=TUNE, init_width=0.7, init=init, order0=order0, table=table,
             weights=WEIGHTS, center=CENTER, samples=torch.stack(samples), widths=torch.stack(widths),
             ticks_used=torch.tensor(ticks_used))

    # mcmc_transform of the reference on two priors
    torch.manual_seed(12)
    mvn = torch.distributions.MultivariateNormal(torch.tensor([1.0, -1.0, 0.5]), torch.diag(torch.tensor([4.0, 0.25, 1.0])))
    box = BoxUniform(-2.0 * torch.ones(3), torch.tensor([3.0, 1.0, 0.0]))
    tr = {}
    for name, prior in (("mvn", mvn), ("box", box)):
        tf = mcmc_transform(prior)
        th = prior.sample((16,))
        u = tf(th)
        tr[name] = dict(theta=th, u=u, lad=tf.log_abs_det_jacobian(th, u))
    g["transform"] = tr
    os.makedirs(OUT, exist_ok=True)
    torch.save(g, os.path.join(OUT, "mcmc_reference.pt"))
    print("wrote", os.path.join(OUT, "mcmc_reference.pt"), "ticks used", min(ticks_used), max(ticks_used))


if __name__ == "__main__":
    main()
