#!/usr/bin/env python
"""C2ST of the trained NSF posterior on the 10-D linear-Gaussian task (north_star gate: <= 0.55), for the
accuracy configuration AND the benchmark configuration (SURVEY.md 8d "C2ST", section 6 last bullet).

Task (tests/mini_sbibm/gaussian_linear.py:30-32, 74-123): prior N(0, 0.1 I), x = theta + sqrt(0.1) eps, analytic
posterior N(x_o / 2, 0.05 I).  Observations idx 1..3: torch.manual_seed(idx) -> prior draw -> one simulator call.
C2ST: sbi.utils.metrics.c2st defaults (random forest, 5-fold, seed 1, z-scored), 10 000 reference vs 10 000 drawn
samples, mean over the three observations.

usage (GPU box):  python tools/c2st_report.py [--out gpurun_out/c2st_report.json] [--quick]
"""
import argparse
import json
import os
import sys
import time
import warnings

import torch
from torch.distributions import MultivariateNormal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DIM = 10


def observation(idx):
    torch.manual_seed(idx)
    prior = MultivariateNormal(torch.zeros(DIM), 0.1 * torch.eye(DIM))
    th = prior.sample((1,))
    return th + (0.1**0.5) * torch.randn(1, DIM)


def run(n_sims, batch, n_draw, max_epochs=None, stop_after=20, tag=""):
    from sbi_amd.inference import NPE
    from sbi_amd.neural_nets import NSFConfig
    from sbi_amd.simulators.linear_gaussian import diagonal_linear_gaussian, true_posterior_linear_gaussian_mvn_prior
    from sbi_amd.utils.metrics import c2st

    torch.manual_seed(0)
    prior = MultivariateNormal(torch.zeros(DIM, device="cuda"), 0.1 * torch.eye(DIM, device="cuda"))
    theta = prior.sample((n_sims,)).cpu()
    x = diagonal_linear_gaussian(theta, std=0.1**0.5)
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator=NSFConfig(), device="cuda", show_progress_bars=False)
    kw = {} if max_epochs is None else {"max_num_epochs": max_epochs}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=batch, stop_after_epochs=stop_after, **kw)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    post = inf.build_posterior()
    scores = []
    for idx in (1, 2, 3):
        x_o = observation(idx)
        ref = true_posterior_linear_gaussian_mvn_prior(x_o, torch.zeros(DIM), 0.1 * torch.eye(DIM), torch.zeros(DIM),
                                                       0.1 * torch.eye(DIM)).sample((n_draw,))
        got = post.sample((n_draw,), x=x_o, show_progress_bars=False).cpu()
        scores.append(c2st(got, ref).item())
    out = {"config": tag, "simulations": n_sims, "training_batch_size": batch, "max_num_epochs": max_epochs,
           "stop_after_epochs": stop_after, "epochs_trained": inf.summary["epochs_trained"][-1],
           "train_seconds": train_s, "best_validation_loss": inf.summary["best_validation_loss"][-1],
           "c2st_per_observation": scores, "c2st_mean": sum(scores) / len(scores), "samples_per_side": n_draw}
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "c2st_report.json"))
    ap.add_argument("--quick", action="store_true", help="1 000 samples per side (as bm_test.py:141-160)")
    a = ap.parse_args()
    nd = 1000 if a.quick else 10000
    res = [
        run(100_000, 1000, nd, tag="accuracy config: 100k sims, batch 1000, sbi defaults (early stopping)"),
        run(100_000, 65536, nd, max_epochs=200, stop_after=10**9,
            tag="benchmark config as timed by bench.py M2: 100k sims, batch 65536, 200 epochs (= 200 steps)"),
        run(100_000, 65536, nd, tag="benchmark batch trained to convergence: 100k sims, batch 65536, early stopping"),
    ]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"task": "10-D linear Gaussian (mini_sbibm gaussian_linear)", "gate": "c2st_mean <= 0.55",
                   "runs": res}, f, indent=1)


if __name__ == "__main__":
    main()
