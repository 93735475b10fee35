#!/usr/bin/env python
"""bench.py -- headline benchmark of the NSF / NPE hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`
prints ONE JSON line on rank 0.  A "step" is one pass of the hot path over one
batch of 65 536 synthetic linear-Gaussian (theta, x) pairs per GPU (weak
scaling): in `train` mode one NPE training step (fused loss fwd+bwd, gradient
all-reduce over RCCL for N>1, fused clip+Adam), in `log_prob` mode one batched
NSF log_prob evaluation.  Inputs are resident in HBM before the timed region.

Workload = BASELINE.json configs[1]: NPE + NSF, theta-dim 10, x-dim 10,
batch 65 536, sbi's default NSF hyper-parameters (98 025 parameters).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = C = 10
BATCH = 65536
F_EVAL = 191_000.0          # dense FLOP per log_prob eval
F_TRAIN = 3 * F_EVAL        # fwd + 2x bwd per training pair (recompute not counted)
PEAK_FP32_MFMA_TFLOPS = 157.3


def make_data(n, device, seed=0):
    """10-D linear-Gaussian task of tests/mini_sbibm/gaussian_linear.py:30-32,101-123:
    prior N(0, 0.1 I), x = theta + sqrt(0.1) eps."""
    g = torch.Generator().manual_seed(seed)
    theta = torch.randn(n, D, generator=g) * (0.1**0.5)
    x = theta + (0.1**0.5) * torch.randn(n, C, generator=g)
    return theta.to(device), x.to(device)


def build_estimator(theta, x, device):
    from sbi_amd.neural_nets.net_builders.flow import build_nsf

    torch.manual_seed(1)
    est = build_nsf(theta.cpu(), x.cpu())
    return est.to(device)


def cpu_baseline(mode, budget_s=12.0):
    """Oracle (= op-for-op restatement of what sbi+nflows execute) timed on host cores."""
    from oracle.nsf_oracle import NSFOracle

    # Intra-op threads: eager PyTorch on (N,145)-sized temporaries stops scaling (and
    # collapses when oversubscribed) well before a 2-socket host's core count.
    cores = min(os.cpu_count() or 1, int(os.environ.get("SBI_AMD_CPU_THREADS", "32")))
    torch.set_num_threads(cores)
    theta, x = make_data(BATCH, "cpu")
    torch.manual_seed(1)
    oracle = NSFOracle(theta, x)
    n = 16384
    th, xx = theta[:n], x[:n]
    if mode == "sample":
        x_o = x[:1]
        with torch.no_grad():
            oracle.sample((n,), x_o)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s or reps < 2:
                oracle.sample((n,), x_o)
                reps += 1
            dt = time.perf_counter() - t0
        return {"value": n * reps / dt, "unit": "draws/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x {n}-draw oracle sample calls, no support check ({dt:.1f} s), "
                          f"torch {torch.__version__} CPU fp32"}
    if mode == "log_prob":
        with torch.no_grad():
            oracle.log_prob(th, xx)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s or reps < 2:
                oracle.log_prob(th, xx)
                reps += 1
            dt = time.perf_counter() - t0
        return {"value": n * reps / dt, "unit": "log_prob evals/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x {n}-row oracle log_prob calls ({dt:.1f} s), torch {torch.__version__} CPU fp32"}
    opt = torch.optim.Adam(oracle.parameters(), lr=5e-4)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s or reps < 2:
        opt.zero_grad()
        loss = oracle.loss(th, xx).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(oracle.parameters(), 5.0)
        opt.step()
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "train pairs/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x {n}-row oracle train steps, pre-batched tensors ({dt:.1f} s), "
                      f"torch {torch.__version__} CPU fp32"}


def fmpe_cpu_baseline(fm, theta, x, budget_s=10.0):
    """Oracle FMPE training step (loss + backward + clip + Adam, eager PyTorch) on host cores."""
    from oracle.fmpe_oracle import FMPEOracle

    cores = min(os.cpu_count() or 1, int(os.environ.get("SBI_AMD_CPU_THREADS", "32")))
    torch.set_num_threads(cores)
    h = fm.net.hyper
    o = FMPEOracle(h.D, h.C, H=h.hidden_features, L=h.num_layers)
    o.load_reference_state_dict({k: v.cpu() for k, v in fm.net.reference_state_dict().items()})
    n = 16384
    th, xx = theta[:n].cpu(), x[:n].cpu()
    opt = torch.optim.Adam(o.parameters(), lr=5e-4)

    def step():
        opt.zero_grad()
        o.loss(th, xx, torch.rand(n), torch.randn(n, h.D)).mean().backward()
        torch.nn.utils.clip_grad_norm_(o.parameters(), 5.0)
        opt.step()

    step()
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        step()
        k += 1
    dt = time.perf_counter() - t0
    return {"value": k * n / dt, "unit": "train pairs/s", "cores": cores, "kind": "port",
            "sample": f"{k} x {n}-row oracle FMPE train steps ({dt:.1f} s), torch {torch.__version__} CPU fp32"}


def fmpe_leg(args, B, rank, world, device, dist, distributed):
    """SURVEY 8f-1 / BASELINE configs[4]: one FMPE training step (default vector-field MLP, theta-dim 50) on
    `--batch` pairs per GPU: draws of t and theta_1, fused CFM loss fwd + bwd, [all-reduce], clip + Adam."""
    from sbi_amd.inference.trainers.fused import FusedFMPEStep
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    DF = 50
    g = torch.Generator().manual_seed(rank)
    th_f = torch.randn(B, DF, generator=g) * (0.1**0.5)
    x_f = (th_f + (0.1**0.5) * torch.randn(B, DF, generator=g)).to(device)
    th_f = th_f.to(device)
    torch.manual_seed(1)
    fm = build_flow_matching_estimator(th_f[:4096].cpu(), x_f[:4096].cpu()).to(device)
    if distributed:
        dist.broadcast(fm.net.flat_params.data, src=0)
    stepper = FusedFMPEStep(fm, lr=5e-4, clip_max_norm=5.0, distributed=distributed)
    wall, dev_ms = timed(lambda: stepper.step(th_f, x_f), args.steps, args.warmup, device, dist)
    h = fm.net.hyper
    H, L, E = h.hidden_features, h.num_layers, h.time_embedding_dim
    f_fwd = 2.0 * (DF * H + DF * H + 2 * H * H + E * H + L * H * H + H * DF)   # dense FLOP per row, forward
    # ODE sampling of the (untrained, random-init) vector field: adaptive Dormand-Prince, every right-hand side one
    # launch of the velocity kernel over all draws; reported as a nested object
    from sbi_amd.inference.posteriors.vector_field_posterior import VectorFieldPosterior

    sample_obj = None
    if rank == 0 and world == 1:   # single-GPU runs only: keeps the N > 1 scaling runs to the collective legs
        with torch.no_grad():   # give the output layer non-zero weights so that the field is not constant
            fm.net.flat_params.add_(0.02 * torch.randn_like(fm.net.flat_params))
        post = VectorFieldPosterior(fm, prior=None, device=str(device))
        n_draw = 65536
        calls = [0]
        real = fm.ode_fn
        fm.ode_fn = lambda *a_, **k_: (calls.__setitem__(0, calls[0] + 1), real(*a_, **k_))[1]
        post.sample((n_draw,), x=x_f[:1])
        torch.cuda.synchronize()
        calls[0] = 0
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            post.sample((n_draw,), x=x_f[:1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        fm.ode_fn = real
        sample_obj = {"metric": "FMPE posterior.sample draws/sec (ODE, atol 1e-6 rtol 1e-5)", "value": n_draw / dt,
                      "unit": "draws/s", "draws_per_call": n_draw, "ms_per_call": dt * 1e3,
                      "velocity_evals_per_call": calls[0] / reps}
    return {
        "posterior_sample": sample_obj,
        "metric": "FMPE train (theta,x)-pairs/sec", "value": B * world * args.steps / wall,
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4] step: FMPE default vector-field MLP (hidden {H}, "
                               f"{L} layers, {h.param_count()} parameters), theta-dim {DF}, x-dim {DF}, "
                               f"batch {B} per GPU, synthetic linear-Gaussian", "parallelism": f"dp{world}"},
        "roofline": roofline(3 * f_fwd, B, args.steps, dev_ms, "fmpe"),
        "_cpu_baseline_fn": lambda: fmpe_cpu_baseline(fm, th_f, x_f)}


def timed(step, steps, warmup, device, dist=None):
    """W warm-up + K timed steps bracketed by barrier + synchronize; returns (wall s [max over ranks], device ms)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)     # HIP events on the stream the kernels are launched on
    t = torch.tensor([wall], device=device, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), dev_ms


# HBM bytes per step at batch 65 536 from the separate rocprofv3 --pmc passes of this same command
# (profiles/r1c_pmc_summary.txt: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, per launch, summed over the step's
# kernels).  bench.py cannot run the profiler on itself, so these are the committed measurements; they are
# only attached when the workload matches the profiled one.
PROFILED_TRAFFIC = {"log_prob": 9.6e6, "train": 1.7e9, "sample": None, "fmpe": 1.83e9}
TRAFFIC_SOURCE = {"fmpe": "profiles/r1e_fmpe_pmc_summary.txt (bytes per step)"}


def roofline(flop_per_unit, units_per_step, steps, dev_ms, kind=None):
    achieved = flop_per_unit * units_per_step * steps / (dev_ms * 1e-3) / 1e12
    traffic = PROFILED_TRAFFIC.get(kind) if units_per_step == BATCH else None
    out = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "device_ms_per_step": dev_ms / steps}
    if traffic is not None:
        out["traffic_source"] = TRAFFIC_SOURCE.get(kind, "profiles/r1c_pmc_summary.txt (bytes per step)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["both", "train", "log_prob", "sample", "atomic", "mcmc", "fmpe"],
                    default=os.environ.get("SBI_AMD_BENCH_MODE", "both"))
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--draws", type=int, default=1_000_000, help="posterior draws per step in the sample leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SBI_AMD_FORCE_DIST=1 exercises the RCCL code path (init, broadcast, all-reduce, barrier) with one rank
    distributed = world > 1 or (os.environ.get("SBI_AMD_FORCE_DIST") == "1" and "RANK" in os.environ)
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    B = args.batch
    theta, x = make_data(B, device, seed=rank)     # per-GPU batch, weak scaling
    est = build_estimator(*make_data(B, "cpu", seed=0), device)
    if distributed:
        dist.broadcast(est.net.flat_params.data, src=0)

    results = {}
    if args.mode in ("both", "log_prob"):
        def lp_step():
            with torch.no_grad():
                est.log_prob(theta, x)

        wall, dev_ms = timed(lp_step, args.steps, args.warmup, device, dist)
        results["log_prob"] = {"value": B * world * args.steps / wall, "unit": "evals/s",
                               "ms_per_step": wall / args.steps * 1e3,
                               "roofline": roofline(F_EVAL, B, args.steps, dev_ms, "log_prob")}
    if args.mode in ("both", "sample"):
        # BASELINE configs[3] (M3): DirectPosterior.sample of 10^6 draws for one x_o, prior support check included
        from torch.distributions import Independent, Normal

        from sbi_amd.inference.posteriors.direct_posterior import DirectPosterior

        prior = Independent(Normal(torch.zeros(D, device=device), (0.1**0.5) * torch.ones(D, device=device)), 1)
        posterior = DirectPosterior(est, prior, device=device)
        x_o = x[:1].clone()
        nd = args.draws

        def sample_step():
            posterior.sample((nd,), x=x_o, max_sampling_batch_size=nd, show_progress_bars=False)

        ssteps = max(1, min(args.steps, 10))
        wall, dev_ms = timed(sample_step, ssteps, min(args.warmup, 2), device, dist)
        results["sample"] = {"value": nd * world * ssteps / wall, "unit": "draws/s", "steps": ssteps,
                             "ms_per_step": wall / ssteps * 1e3, "roofline": roofline(F_EVAL, nd, ssteps, dev_ms)}
    if args.mode == "mcmc":
        # MCMCPosterior (slice_np_vectorized on the device) over the NSF potential, one x_o
        from torch.distributions import Independent, Normal

        from sbi_amd.inference.posteriors.mcmc_posterior import MCMCPosterior
        from sbi_amd.inference.potentials.posterior_based_potential import posterior_estimator_based_potential
        from sbi_amd.utils.sbiutils import mcmc_transform

        prior = Independent(Normal(torch.zeros(D, device=device), (0.1**0.5) * torch.ones(D, device=device)), 1)
        potential_fn, _ = posterior_estimator_based_potential(est, prior, x_o=None)
        chains = int(os.environ.get("SBI_AMD_MCMC_CHAINS", "4096"))
        post = MCMCPosterior(potential_fn, prior, mcmc_transform(prior, device=device), num_chains=chains, thin=1,
                             warmup_steps=10, init_strategy="resample",
                             init_strategy_parameters=dict(num_candidate_samples=64), device=str(device))
        post.set_default_x(x[:1].clone())
        nd = chains * 10
        post.sample((nd,), show_progress_bars=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        post.sample((nd,), show_progress_bars=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ticks = post.posterior_sampler.num_ticks
        if rank == 0:
            print(json.dumps({"metric": "MCMCPosterior slice_np_vectorized samples/sec", "value": nd / dt,
                              "unit": "samples/s", "n_gpus": 1, "chains": chains, "ticks": ticks,
                              "us_per_tick": dt / ticks * 1e6, "log_prob_evals_per_s": ticks * chains / dt,
                              "config": {"workload": f"{chains} chains x {nd // chains} kept sweeps (+10 warm-up, "
                                                     f"+50 width-tuning sweeps), theta-dim {D}, one x_o"}}))
        return
    if args.mode == "fmpe":
        out = fmpe_leg(args, B, rank, world, device, dist, distributed)
        if rank == 0:
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = out.pop("_cpu_baseline_fn")()
            out.pop("_cpu_baseline_fn", None)
            print(json.dumps(out))
        if distributed:
            dist.destroy_process_group()
        return
    if args.mode == "atomic":
        # one multi-round NPE-C step (atomic proposal-posterior loss, 10 atoms) on `--batch` pairs
        from torch.distributions import Independent, Normal

        from sbi_amd.inference.trainers.fused import FusedTrainStep

        prior = Independent(Normal(torch.zeros(D, device=device), (0.1**0.5) * torch.ones(D, device=device)), 1)
        masks = torch.zeros(B, 1, dtype=torch.bool, device=device)
        stepper = FusedTrainStep(est, lr=5e-4, clip_max_norm=5.0, distributed=distributed)
        A = 10
        wall, dev_ms = timed(lambda: stepper.atomic_step(theta, x, masks, prior, A), args.steps, args.warmup, device,
                             dist)
        if rank == 0:
            print(json.dumps({
                "metric": "NPE-C atomic-loss train (theta,x)-pairs/sec", "value": B * world * args.steps / wall,
                "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"multi-round NPE-C step, {A} atoms, batch {B} per GPU = {A * B} log_prob "
                                       f"rows forward + backward, theta-dim {D}", "parallelism": f"dp{world}"},
                "roofline": roofline(F_TRAIN, A * B, args.steps, dev_ms)}))
        if distributed:
            dist.destroy_process_group()
        return
    if args.mode in ("both", "train"):
        from sbi_amd.inference.trainers.fused import FusedTrainStep

        stepper = FusedTrainStep(est, lr=5e-4, clip_max_norm=5.0, distributed=distributed)
        wall, dev_ms = timed(lambda: stepper.step(theta, x), args.steps, args.warmup, device, dist)
        results["train"] = {"value": B * world * args.steps / wall, "unit": "pairs/s",
                            "ms_per_step": wall / args.steps * 1e3,
                            "roofline": roofline(F_TRAIN, B, args.steps, dev_ms, "train")}

    fm_out = fmpe_leg(args, B, rank, world, device, dist, distributed) if args.mode == "both" else None
    if rank == 0:
        head = "train" if "train" in results else ("log_prob" if "log_prob" in results else "sample")
        r = results[head]
        out = {
            "metric": {"train": "NPE train (theta,x)-pairs/sec", "log_prob": "NSF log_prob evals/sec",
                       "sample": "DirectPosterior.sample draws/sec"}[head],
            "value": r["value"], "unit": r["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: NPE + NSF theta-dim {D}, x-dim {C}, batch {B} per GPU, "
                                   f"synthetic linear-Gaussian; step = one fused NPE training step"
                       if head == "train" else
                       f"NSF log_prob, theta-dim {D}, x-dim {C}, batch {B} per GPU", "parallelism": f"dp{world}"},
            # whole step (forward + T backward launches + reduce + clip/Adam) against dense fp32 MFMA;
            # per-kernel durations: profiles/*kernel_stats.csv
            "roofline": r["roofline"],
        }
        if "log_prob" in results and head == "train":
            lp = results["log_prob"]
            out["log_prob"] = {"metric": "NSF log_prob evals/sec", "value": lp["value"], "unit": lp["unit"],
                               "ms_per_step": lp["ms_per_step"], "roofline": lp["roofline"]}
        if "sample" in results and head != "sample":
            sp = results["sample"]
            out["posterior_sample"] = {"metric": "DirectPosterior.sample draws/sec", "value": sp["value"],
                                       "unit": sp["unit"], "draws_per_step": args.draws, "steps": sp["steps"],
                                       "ms_per_step": sp["ms_per_step"], "roofline": sp["roofline"]}
        if fm_out is not None:
            out["fmpe_train"] = {k: fm_out[k] for k in ("metric", "value", "unit", "ms_per_step", "roofline",
                                                        "posterior_sample")}
            out["fmpe_train"]["workload"] = fm_out["config"]["workload"]
            if world == 1 and not args.no_cpu_baseline:
                out["fmpe_train"]["cpu_baseline"] = fm_out["_cpu_baseline_fn"]()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(head)
            if "posterior_sample" in out:
                out["posterior_sample"]["cpu_baseline"] = cpu_baseline("sample", budget_s=6.0)
            if "log_prob" in results and head == "train":
                out["log_prob"]["cpu_baseline"] = cpu_baseline("log_prob", budget_s=8.0)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
