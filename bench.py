#!/usr/bin/env python
"""bench.py -- headline benchmark of the NSF / NPE hot path on MI355X.

Contract (task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.

* N ranks, one per GPU, RCCL backend.  The driver launches N > 1 as `python -m torch.distributed.run
  --nproc-per-node N ... bench.py --gpus N ...` (RANK / WORLD_SIZE in the environment); a bare
  `python bench.py --gpus N` with N > 1 re-launches ITSELF the same way (`launch_ranks`).  A world size that
  differs from `--gpus`, or fewer visible devices than ranks, is an error -- never a silent 1-GPU run.
* A "step" of the headline `train` leg is one pass of sbi's training inner loop (trainers/base.py:1150-1193) over
  one batch: device-side gather of the batch from the resident simulations in a fresh pseudo-random order (the
  role of SubsetRandomSampler; one launch, sbi_amd/utils/shuffle.py) -> weight re-pack -> fused loss forward + backward -> [ONE all-reduce of the flat
  98 025-float gradient over RCCL] -> fused global-norm clip + Adam.  Inputs are resident in HBM.
* Timing of every leg: [~150 ms of UNTIMED steps of the same leg, `preheat()`: a device that was idle runs its first
  ~40 ms slower, and with W = 5, K = 20 the timed region would measure that ramp] -> W warm-up steps -> barrier +
  synchronize -> EXACTLY K timed steps -> barrier + synchronize; max over ranks.  The JSON names the preheat.
* `--scaling weak` (default): 65 536 pairs per GPU per step.  `--scaling strong`: 65 536 pairs per step split N
  ways.  At N > 1 the weak line also carries the strong number as a nested object.

Workload = BASELINE.json configs[1]: NPE + NSF, theta-dim 10, x-dim 10, 100 000 simulations, batch 65 536, sbi's
default NSF hyper-parameters (98 025 parameters).  Nested objects: `log_prob` (M1, paired x) and
`log_prob_broadcast_x` (M1, one x_o), `posterior_sample` (M3, 10^6 draws), `npe_train` (M2 exactly as SURVEY 8d
defines it: `NPE.train()` on 100 000 simulations, batch 65 536, validation pass and early-stopping bookkeeping
included), `fmpe_train` (configs[4] step).
"""

from __future__ import annotations

import argparse
import gc
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = C = 10
BATCH = 65536
N_SIMS = 100_000
F_EVAL = 191_000.0          # dense FLOP per log_prob eval
F_TRAIN = 3 * F_EVAL        # fwd + 2x bwd per training pair (recompute not counted)
PEAK_FP32_MFMA_TFLOPS = 157.3


# ----------------------------------------------------------------------------------------- rank launching
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, script: str, argv, require_gpus: bool = True, extra_env=None) -> int:
    """Re-launch `script argv` as n ranks of ONE node through torch.distributed.run (127.0.0.1 rendezvous).
    Returns the launcher's exit code.  Raises when the node has fewer GPUs than ranks."""
    if require_gpus:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit(f"bench.py: --gpus {n} requested but only {have} ROCm device(s) are visible; "
                             "refusing to run fewer ranks than asked for")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    return subprocess.call(cmd, env=env)


def rccl_one_rank_leg(args, plain):
    """The train leg once more as rank 0 of a ONE-rank `nccl` (= RCCL) process group, self-launched through
    torch.distributed.run exactly as the driver launches N > 1: communicator init, parameter broadcast, the
    all-reduce of the flat gradient between backward and clip + Adam, barriers -- the data-parallel code path on the
    one GPU a 1-GPU box has.  `plain` = this process's own train result (no process group) for the ratio."""
    env = dict(os.environ)
    env.update({"SBI_AMD_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                "OMP_NUM_THREADS": env.get("OMP_NUM_THREADS", "4")})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), "--gpus", "1", "--mode", "train",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch), "--no-cpu-baseline",
           "--no-rccl-leg"]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = next(l for l in reversed(p.stdout.splitlines()) if l.startswith("{"))
        j = json.loads(line)
    except Exception as e:   # noqa: BLE001 -- the headline must not die with the side leg
        return {"error": f"{type(e).__name__}: {e}"}
    return {"metric": "NPE train (theta,x)-pairs/sec as rank 0 of a 1-rank RCCL group", "value": j["value"],
            "unit": j["unit"], "ms_per_step": j["ms_per_step"],
            "fused_step_device_ms": j["roofline"]["device_ms_per_step"],
            "rccl_ranks": j["config"]["rccl_ranks"],
            "allreduce_us_98025_floats": j["config"]["rccl_allreduce_us_98025_floats"],
            "ms_per_step_vs_no_process_group": j["ms_per_step"] / plain["ms_per_step"],
            "launched_as": "python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 ... bench.py --gpus 1 "
                           "--mode train (SBI_AMD_FORCE_DIST=1)"}


# ----------------------------------------------------------------------------------------- data / model
def make_data(n, device, seed=0, dim=D):
    """Linear-Gaussian task of tests/mini_sbibm/gaussian_linear.py:30-32,101-123: prior N(0, 0.1 I),
    x = theta + sqrt(0.1) eps."""
    g = torch.Generator().manual_seed(seed)
    theta = torch.randn(n, dim, generator=g) * (0.1**0.5)
    x = theta + (0.1**0.5) * torch.randn(n, dim, generator=g)
    return theta.to(device), x.to(device)


def build_estimator(theta, x, device):
    from sbi_amd.neural_nets.net_builders.flow import build_nsf

    torch.manual_seed(1)
    est = build_nsf(theta.cpu(), x.cpu())
    return est.to(device)


# ----------------------------------------------------------------------------------------- CPU baselines
def _cpu_cores():
    # Intra-op threads: eager PyTorch on (N,145)-sized temporaries stops scaling (and collapses when
    # oversubscribed) well before a 2-socket host's core count.
    cores = min(os.cpu_count() or 1, int(os.environ.get("SBI_AMD_CPU_THREADS", "32")))
    torch.set_num_threads(cores)
    return cores


def cpu_baseline(mode, budget_s=12.0):
    """Oracle (= op-for-op restatement of what sbi + nflows execute) timed on the host cores, at the SAME batch
    of 65 536 rows as the GPU legs (bounded by wall time: at least 2 repetitions)."""
    from oracle.nsf_oracle import NSFOracle

    cores = _cpu_cores()
    theta, x = make_data(BATCH, "cpu")
    torch.manual_seed(1)
    oracle = NSFOracle(theta, x)
    n = BATCH
    th, xx = theta[:n], x[:n]
    tag = f"torch {torch.__version__} CPU fp32, {cores} intra-op threads"
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
                "sample": f"{reps} x {n}-draw oracle sample calls, no support check ({dt:.1f} s), {tag}"}
    if mode in ("log_prob", "log_prob_broadcast_x"):
        xb = x[:1].expand(n, C) if mode == "log_prob_broadcast_x" else xx   # nflows materialises the repeat
        with torch.no_grad():
            oracle.log_prob(th, xb)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s or reps < 2:
                oracle.log_prob(th, xb.contiguous() if mode == "log_prob_broadcast_x" else xb)
                reps += 1
            dt = time.perf_counter() - t0
        return {"value": n * reps / dt, "unit": "log_prob evals/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x {n}-row oracle log_prob calls ({dt:.1f} s), {tag}"}
    opt = torch.optim.Adam(oracle.parameters(), lr=5e-4)
    if mode == "train_dataloader":
        # the reference's REAL loader path (trainers/base.py:525-561): TensorDataset(theta, x, prior_masks) behind a
        # DataLoader with a SubsetRandomSampler over the 90 000-row training split, batch 65 536, drop_last -- i.e.
        # per-ELEMENT __getitem__ + default_collate of 65 536 triples per step, then the same oracle step
        from torch.utils import data

        th_all, x_all = make_data(N_SIMS, "cpu")
        ds = data.TensorDataset(th_all, x_all, torch.ones(N_SIMS, 1))
        train_idx = torch.randperm(N_SIMS)[: int(0.9 * N_SIMS)]
        loader = data.DataLoader(ds, batch_size=min(BATCH, train_idx.numel()), drop_last=True,
                                 sampler=data.SubsetRandomSampler(train_idx.tolist()))
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s or reps < 2:
            for tb, xb, _ in loader:           # one 65 536-row batch per epoch (90 000 // 65 536 with drop_last)
                opt.zero_grad()
                oracle.loss(tb, xb).mean().backward()
                torch.nn.utils.clip_grad_norm_(oracle.parameters(), 5.0)
                opt.step()
                reps += 1
        dt = time.perf_counter() - t0
        return {"value": n * reps / dt, "unit": "train pairs/s", "cores": cores, "kind": "port",
                "sample": f"{reps} epochs of sbi's loader path (TensorDataset + SubsetRandomSampler + default_collate of "
                          f"{n} elements per step, trainers/base.py:525-561) + the oracle train step ({dt:.1f} s), {tag}"}
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
            "sample": f"{reps} x {n}-row oracle train steps (loss, backward, clip, Adam; pre-batched tensors, no "
                      f"DataLoader) ({dt:.1f} s), {tag}"}


def fmpe_cpu_baseline(fm, theta, x, budget_s=10.0):
    """Oracle FMPE training step (loss + backward + clip + Adam, eager PyTorch) on host cores."""
    from oracle.fmpe_oracle import FMPEOracle

    cores = _cpu_cores()
    h = fm.net.hyper
    o = FMPEOracle(h.D, h.C, H=h.hidden_features, L=h.num_layers)
    o.load_reference_state_dict({k: v.cpu() for k, v in fm.net.reference_state_dict().items()})
    n = min(BATCH, theta.shape[0])
    th, xx = theta[:n].cpu(), x[:n].cpu()
    opt = torch.optim.Adam(o.parameters(), lr=5e-4)

    def step():
        opt.zero_grad()
        o.loss(th, xx, torch.rand(n), torch.randn(n, h.D)).mean().backward()
        torch.nn.utils.clip_grad_norm_(o.parameters(), 5.0)
        opt.step()

    step()
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s or k < 2:
        step()
        k += 1
    dt = time.perf_counter() - t0
    return {"value": k * n / dt, "unit": "train pairs/s", "cores": cores, "kind": "port",
            "sample": f"{k} x {n}-row oracle FMPE train steps ({dt:.1f} s), torch {torch.__version__} CPU fp32"}


# ----------------------------------------------------------------------------------------- timing / roofline
LAST_TIMED: dict = {}     # host-side enqueue statistics of the most recent timed() call


PREHEAT_MS = float(os.environ.get("SBI_AMD_BENCH_PREHEAT_MS", "150"))


def preheat(step, device, dist=None) -> int:
    """Untimed: keep the device busy with the leg's own step for ~PREHEAT_MS before the W warm-up steps.  After >= 10 ms
    of idleness the device runs its next ~20 ms of work up to 15 % slower (clock ramp; measured per step by
    tools/diag/clock_ramp.py, `profiles/r6m_clock_ramp.txt`: 65 536-row step 0.835 / 0.80 / 0.76 ms over steps 0-4 /
    5-9 / 10-19 after a 10 ... 500 ms pause, 0.722 ms from step ~20 on; no effect after a 2 ms pause).  With the
    driver's W = 5, K = 20 -- and a 50 ms `gc.collect()` between warm-up and timed steps, as this file had it until
    round 6 -- the timed region sat entirely inside that ramp: the "slow box" of earlier profile sets.  The number of
    steps comes from a 3-step probe, max over ranks, so every rank of a data-parallel run does the same count (each
    step holds a collective).  SBI_AMD_BENCH_PREHEAT_MS=0 switches it off."""
    if PREHEAT_MS <= 0:
        return 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    if dist is not None:
        t = torch.tensor([ms], dtype=torch.float64, device=device if "nccl" in str(dist.get_backend()).lower() else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    n = int(min(1000, max(0, -(-PREHEAT_MS // max(ms, 1e-3)) - 3)))
    for _ in range(n):
        step()
    return n + 3


def timed(step, steps, warmup, device, dist=None):
    """[untimed preheat, see above] + W warm-up + K timed steps bracketed by barrier + synchronize; returns
    (wall s [max over ranks], device ms)."""
    # The cyclic garbage collector stays out of the timed region (as `timeit` does): a generation-2 pass over the
    # objects earlier legs left behind is tens of milliseconds of HOST time that would be charged to K device steps
    # (SBI_AMD_BENCH_KEEP_GC=1 keeps it on, for A/B).  It runs BEFORE the warm-up: nothing that idles the device for
    # milliseconds may sit between the last warm-up step and the first timed one (see preheat()).
    gc_on = gc.isenabled()
    if os.environ.get("SBI_AMD_BENCH_KEEP_GC") != "1":
        gc.collect()
        gc.disable()
    n_pre = preheat(step, device, dist)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = []
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        h0 = time.perf_counter()
        step()
        host.append(time.perf_counter() - h0)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if gc_on:
        gc.enable()
    LAST_TIMED.clear()
    LAST_TIMED.update(preheat_steps=n_pre, host_enqueue_ms_max=max(host) * 1e3, host_enqueue_ms_argmax=host.index(max(host)),
                      host_enqueue_ms_median=sorted(host)[len(host) // 2] * 1e3)
    dev_ms = ev0.elapsed_time(ev1)     # HIP events on the stream the kernels are launched on
    t = torch.tensor([wall], dtype=torch.float64, device=device if dist is None or "nccl" in str(dist.get_backend()).lower() else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), dev_ms


def load_traffic():
    """HBM bytes per step from the newest committed `profiles/*traffic.json`, which `tools/profile_round.sh` writes
    from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command (gfx950
    correction of the guide applied: FETCH_SIZE x 2).  bench.py cannot profile itself, so the line carries the
    committed measurement together with its source file and the commit it was measured at."""
    # (newest = last in name order, profiles/<round tag>_traffic.json: file times do not survive a checkout)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            t = json.load(f)
        t["_file"] = os.path.relpath(files[-1], ROOT)
        return t
    except (OSError, ValueError):
        return None


_TRAFFIC = None


def roofline(flop_per_unit, units_per_step, steps, dev_ms, kind=None):
    global _TRAFFIC
    achieved = flop_per_unit * units_per_step * steps / (dev_ms * 1e-3) / 1e12
    out = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None, "device_ms_per_step": dev_ms / steps}
    if kind is not None and units_per_step == BATCH:
        if _TRAFFIC is None:
            _TRAFFIC = load_traffic() or {}
        per_step = (_TRAFFIC.get("bytes_per_step") or {}).get(kind)
        if per_step is not None:
            out["traffic"] = per_step
            out["traffic_source"] = f"{_TRAFFIC['_file']} (rocprofv3 --pmc passes at commit " \
                                    f"{_TRAFFIC.get('commit', '?')}; bytes per step at batch {BATCH})"
    return out


# ----------------------------------------------------------------------------------------- legs
class TrainLeg:
    """The NPE inner loop on resident simulations: per step the batch's rows gathered in a fresh pseudo-random order
    (one launch of the device-side sampler `NPE.train` uses, sbi_amd/utils/shuffle.py: every step is a new epoch's
    order over the resident training rows) and the fused step (what `NPE.train` does per minibatch)."""

    def __init__(self, est, theta_all, x_all, batch, distributed, global_batch):
        from sbi_amd.inference.trainers.fused import FusedTrainStep

        self.theta_all, self.x_all, self.batch = theta_all, x_all, batch
        self.global_batch = global_batch
        self.stepper = FusedTrainStep(est, lr=5e-4, clip_max_norm=5.0, distributed=distributed)
        self.events = []          # (start, end) HIP events around the fused step alone (no sampler / gather)
        from sbi_amd.utils.shuffle import ShuffledGather

        self.sampler = ShuffledGather(theta_all, x_all, None, seed=20260924)
        self.calls = 0

    def __call__(self):
        th, xx = self.sampler.batch(self.calls, 0, self.batch)
        self.calls += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.stepper.step(th, xx, global_batch=self.global_batch)
        e1.record()
        self.events.append((e0, e1))

    def fused_ms(self, last: int) -> float:
        """Sum of the fused-step device times of the last `last` calls (after the final synchronize)."""
        return sum(a.elapsed_time(b) for a, b in self.events[-last:])


def npe_train_leg(device, rank, world, epochs):
    """M2 as SURVEY 8d defines it: `NPE.train()` on 100 000 simulations with training_batch_size 65 536 (90 000 /
    10 000 split => one 65 536-row training step + one 10 000-row validation step per epoch), fixed number of
    epochs, early stopping disabled by a large `stop_after_epochs`; plus the dense variant (10 x 65 536 / 0.9
    simulations => 10 training steps per epoch).  Wall time of the whole call: network construction, z-scoring,
    per-epoch permutations, validation, best-weights bookkeeping and the per-epoch host read all included."""
    import warnings

    from torch.distributions import Independent, Normal

    from sbi_amd.inference import NPE

    def run(n_sims, batch, ep):
        prior = Independent(Normal(torch.zeros(D, device=device), (0.1**0.5) * torch.ones(D, device=device)), 1)
        theta, x = make_data(n_sims, "cpu", seed=0)
        torch.manual_seed(1)
        inf = NPE(prior=prior, density_estimator="nsf", device=str(device), show_progress_bars=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            inf.append_simulations(theta, x)
            inf.train(training_batch_size=batch, max_num_epochs=2, stop_after_epochs=10**9)   # warm-up (build, alloc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            inf.train(training_batch_size=batch, max_num_epochs=ep + 2, stop_after_epochs=10**9,
                      resume_training=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        n_train = int(0.9 * n_sims)
        steps = (n_train // batch) * ep
        bv = min(batch, n_sims - n_train)
        return {"value": batch * steps / dt, "unit": "pairs/s", "train_steps": steps, "epochs": ep,
                "ms_per_epoch": dt / ep * 1e3, "simulations": n_sims, "training_batch_size": batch,
                "validation_rows_per_epoch": ((n_sims - n_train) // bv) * bv,
                "final_validation_loss": inf.summary["validation_loss"][-1]}

    head = run(N_SIMS, BATCH, epochs)
    dense = run(728_200, BATCH, max(2, epochs // 10))
    # sbi's default training_batch_size on the same simulations: 450 steps + 50 validation batches per epoch
    small = run(N_SIMS, 200, 6)
    return {"metric": "NPE.train() (theta,x)-pairs/sec (M2, SURVEY 8d)", **head, "dense_epochs": dense,
            "batch_200": small, "n_gpus": world}


def fmpe_leg(args, B, rank, world, device, dist, distributed, global_batch):
    """SURVEY 8f-1 / BASELINE configs[4]: one FMPE training step (default vector-field MLP, theta-dim 50) on
    B pairs per GPU: draws of t and theta_1, fused CFM loss fwd + bwd, [all-reduce], clip + Adam."""
    from sbi_amd.inference.trainers.fused import FusedFMPEStep
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    DF = 50
    th_f, x_f = make_data(B, device, seed=rank, dim=DF)
    torch.manual_seed(1)
    fm = build_flow_matching_estimator(th_f[:4096].cpu(), x_f[:4096].cpu()).to(device)
    if distributed:
        from sbi_amd.utils.collectives import broadcast_from_rank0

        broadcast_from_rank0(dist, fm.net.flat_params.data)
    stepper = FusedFMPEStep(fm, lr=5e-4, clip_max_norm=5.0, distributed=distributed)
    wall, dev_ms = timed(lambda: stepper.step(th_f, x_f, global_batch=global_batch), args.steps, args.warmup, device,
                         dist)
    host_stats = dict(LAST_TIMED)
    h = fm.net.hyper
    H, L, E = h.hidden_features, h.num_layers, h.time_embedding_dim
    f_fwd = 2.0 * (DF * H + DF * H + 2 * H * H + E * H + L * H * H + H * DF)   # dense FLOP per row, forward
    # ODE sampling of the (untrained, random-init) vector field: adaptive Dormand-Prince, every right-hand side one
    # launch of the velocity kernel over all draws; reported as a nested object
    from sbi_amd.inference.posteriors.vector_field_posterior import VectorFieldPosterior

    sample_obj = None
    if rank == 0 and world == 1 and not args.skip_sampling:   # single-GPU runs only
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
        # log-density of the same flow (VectorFieldPosterior.log_prob): the augmented ODE with the exact Jacobian
        # trace, every right-hand side one launch of the velocity + divergence kernel over all rows
        n_lp = 8192
        th_lp = post.sample((n_lp,), x=x_f[:1])
        real2 = fm.ode_fn_and_divergence
        calls[0] = 0
        fm.ode_fn_and_divergence = lambda *a_, **k_: (calls.__setitem__(0, calls[0] + 1), real2(*a_, **k_))[1]
        post.log_prob(th_lp, x=x_f[:1])
        torch.cuda.synchronize()
        calls[0] = 0
        t0 = time.perf_counter()
        for _ in range(reps):
            lp_ = post.log_prob(th_lp, x=x_f[:1])
        torch.cuda.synchronize()
        dt_lp = (time.perf_counter() - t0) / reps
        fm.ode_fn_and_divergence = real2
        sample_obj["log_prob"] = {"metric": "FMPE posterior.log_prob evals/sec (augmented ODE, exact trace, atol 1e-6 "
                                            "rtol 1e-5)", "value": n_lp / dt_lp, "unit": "evals/s", "rows": n_lp,
                                  "theta_dim": DF, "ms_per_call": dt_lp * 1e3,
                                  "rhs_evals_per_call": calls[0] / reps, "finite": bool(torch.isfinite(lp_).all())}
    # BASELINE configs[4] through the trainer itself: FMPE.train() on 10^6 simulations (theta-dim 50), batch 65 536,
    # validation at 10 fixed times, EMA / early-stopping bookkeeping and the per-epoch host read included
    loop_obj = None
    if args.mode == "fmpe" and not args.skip_sampling:
        import warnings

        from sbi_amd.inference import FMPE

        n_sims, ep = 1_000_000, 4
        th_all, x_all = make_data(n_sims, "cpu", seed=7, dim=DF)
        torch.manual_seed(1)
        trainer = FMPE(prior=None, device=str(device), show_progress_bars=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            trainer.append_simulations(th_all, x_all)
            trainer.train(training_batch_size=B, max_num_epochs=1, stop_after_epochs=10**9)     # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trainer.train(training_batch_size=B, max_num_epochs=1 + ep, stop_after_epochs=10**9, resume_training=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        steps_ep = int(0.9 * n_sims) // B
        loop_obj = {"metric": "FMPE.train() (theta,x)-pairs/sec, 10^6 simulations", "value": B * steps_ep * ep / dt,
                    "unit": "pairs/s", "epochs": ep, "train_steps": steps_ep * ep, "ms_per_epoch": dt / ep * 1e3,
                    "validation_rows_per_epoch": ((n_sims - int(0.9 * n_sims)) // B) * B * 10, "n_gpus": world}
    return {
        "posterior_sample": sample_obj, "train_loop": loop_obj,
        "metric": "FMPE train (theta,x)-pairs/sec", "value": B * world * args.steps / wall,
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4] step: FMPE default vector-field MLP (hidden {H}, "
                               f"{L} layers, {h.param_count()} parameters), theta-dim {DF}, x-dim {DF}, "
                               f"batch {B} per GPU, synthetic linear-Gaussian", "parallelism": f"dp{world}"},
        "roofline": roofline(3 * f_fwd, B, args.steps, dev_ms, "fmpe"), "host": host_stats,
        "_cpu_baseline_fn": lambda: fmpe_cpu_baseline(fm, th_f, x_f)}


def mcmc_leg(est, x, device, with_cpu=False):
    """MCMCPosterior (slice_np_vectorized on the device, sbi/samplers/mcmc/slice_numpy.py:353-587) over the
    NSF potential, one x_o: the persistent sampler kernel (sbi_amd_mcmc_slice_run) runs 64 ticks of every chain per launch
    -- a workgroup owns 16 chains and alternates their log-density (cooperative forward pass) with their tick."""
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
    out = {"metric": "MCMCPosterior slice_np_vectorized samples/sec", "value": nd / dt,
           "unit": "samples/s", "n_gpus": 1, "chains": chains, "ticks": ticks,
           "us_per_tick": dt / ticks * 1e6, "log_prob_evals_per_s": ticks * chains / dt,
           "config": {"workload": f"{chains} chains x {nd // chains} kept sweeps (+10 warm-up, "
                                  f"+50 width-tuning sweeps), theta-dim {D}, one x_o"}}
    if with_cpu:
        # what bounds the reference's sampler on the host: one potential evaluation of all chains per tick (the oracle's
        # log_prob at `chains` rows, x_o repeated as nflows does) -- its slice bookkeeping is numpy and comes on top
        from oracle.nsf_oracle import NSFOracle

        th_c, x_c = make_data(BATCH, "cpu")
        torch.manual_seed(1)
        oracle = NSFOracle(th_c, x_c)
        xb = x_c[:1].expand(chains, C).contiguous()
        with torch.no_grad():
            oracle.log_prob(th_c[:chains], xb)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 4.0 or reps < 2:
                oracle.log_prob(th_c[:chains], xb)
                reps += 1
            dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": chains * reps / dtc, "unit": "log_prob evals/s", "cores": _cpu_cores(),
                               "kind": "port", "compare_with": "log_prob_evals_per_s",
                               "sample": f"{reps} oracle log_prob calls of {chains} rows (one tick's potential "
                                         f"evaluation; the host-side slice bookkeeping is not included) ({dtc:.1f} s)"}
    return out


def atomic_leg(est, theta, x, B, GB, steps, warmup, device, dist, distributed, world, scaling, with_cpu=False):
    """one multi-round NPE-C step (atomic proposal-posterior loss, sbi/inference/trainers/npe/npe_c.py:356-440,
    10 atoms) on B pairs: A x B log_prob rows forward + backward on one stash, softmax weights on the device."""
    from torch.distributions import Independent, Normal

    from sbi_amd.inference.trainers.fused import FusedTrainStep

    prior = Independent(Normal(torch.zeros(D, device=device), (0.1**0.5) * torch.ones(D, device=device)), 1)
    masks = torch.zeros(B, 1, dtype=torch.bool, device=device)
    stepper = FusedTrainStep(est, lr=5e-4, clip_max_norm=5.0, distributed=distributed)
    A = 10
    wall, dev_ms = timed(lambda: stepper.atomic_step(theta, x, masks, prior, A), steps, warmup, device, dist)
    out = {"metric": "NPE-C atomic-loss train (theta,x)-pairs/sec", "value": GB * steps / wall,
           "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": scaling,
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"multi-round NPE-C step, {A} atoms, batch {B} per GPU = {A * B} log_prob "
                                  f"rows forward + backward, theta-dim {D}", "parallelism": f"dp{world}"},
           "roofline": roofline(F_TRAIN, A * B, steps, dev_ms)}
    if with_cpu:
        from oracle.nsf_oracle import NSFOracle
        from sbi_amd.inference.trainers.npe.atomic import log_prob_proposal_posterior_atomic

        th_c, x_c = make_data(BATCH, "cpu")
        torch.manual_seed(1)
        oracle = NSFOracle(th_c, x_c)
        bc = 2048        # bounded sample: 2 048 pairs x 10 atoms per step
        prior_c = Independent(Normal(torch.zeros(D), (0.1**0.5) * torch.ones(D)), 1)
        opt = torch.optim.Adam(oracle.parameters(), lr=5e-4)
        mk = torch.zeros(bc, 1, dtype=torch.bool)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 5.0 or reps < 2:
            opt.zero_grad()
            (-log_prob_proposal_posterior_atomic(oracle, prior_c, th_c[:bc], x_c[:bc], mk, A, False)).mean().backward()
            torch.nn.utils.clip_grad_norm_(oracle.parameters(), 5.0)
            opt.step()
            reps += 1
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": bc * reps / dtc, "unit": "pairs/s", "cores": _cpu_cores(), "kind": "port",
                               "sample": f"{reps} atomic-loss steps of {bc} pairs x {A} atoms through the oracle "
                                         f"(autograd, clip, Adam) ({dtc:.1f} s)"}
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["both", "train", "log_prob", "log_prob_broadcast", "sample", "atomic", "mcmc", "fmpe",
                                      "npe_train", "maf", "zuko"],
                    default=os.environ.get("SBI_AMD_BENCH_MODE", "both"))
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch pairs per GPU per step; strong: --batch pairs per step split over the GPUs")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--draws", type=int, default=1_000_000, help="posterior draws per step in the sample leg")
    ap.add_argument("--npe-epochs", type=int, default=200, help="epochs of the NPE.train() (M2) leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-sampling", action="store_true", help="fmpe mode: no ODE-sampling leg (profiling passes)")
    ap.add_argument("--no-small-batch", action="store_true",
                    help="train leg without the batch-200 / 8 192 object (traffic passes: bytes per step of the headline)")
    ap.add_argument("--no-rccl-leg", action="store_true",
                    help="N = 1: skip the self-launched 1-rank RCCL run of the train leg (`rccl_1rank` object)")
    args = ap.parse_args(argv)

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: become one (one process per GPU, RCCL)
        raise SystemExit(launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:] if argv is None else argv,
                                      require_gpus=os.environ.get("SBI_AMD_BENCH_SHARE_GPU") != "1"))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # SBI_AMD_FORCE_DIST=1 exercises the RCCL code path (init, broadcast, all-reduce, barrier) with one rank
    distributed = world > 1 or (os.environ.get("SBI_AMD_FORCE_DIST") == "1" and "RANK" in os.environ)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (there is no CPU path)")
    # SBI_AMD_BENCH_SHARE_GPU=1 (tests, 1-GPU boxes): every rank sits on device 0 and the group is `gloo` (RCCL refuses
    # two ranks on one device; device buffers are staged through the host by sbi_amd/utils/collectives.py).  The DP
    # code path -- windows of one global batch, 1/global_batch weighting, all-reduce, max-over-ranks timing -- is the
    # product's; the NUMBER is not a scaling measurement and the line says so (`config.shared_gpu`).
    share_gpu = os.environ.get("SBI_AMD_BENCH_SHARE_GPU") == "1" and world > 1
    if share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    rccl_world = 1
    allreduce_us = None
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from sbi_amd.utils.collectives import all_reduce_sum, broadcast_from_rank0

        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        probe = torch.ones(1, device=device)
        all_reduce_sum(dist, probe)
        rccl_world = int(probe.item())          # the number of ranks RCCL actually reduced over
        if rccl_world != world:
            raise SystemExit(f"bench.py: RCCL all-reduce saw {rccl_world} ranks, expected {world}")
        # the collective of one training step in isolation: all-reduce of a flat 98 025-float gradient buffer
        gbuf = torch.zeros(98_025, device=device)
        for _ in range(10):
            all_reduce_sum(dist, gbuf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            all_reduce_sum(dist, gbuf)
        e1.record()
        torch.cuda.synchronize()
        allreduce_us = e0.elapsed_time(e1) / 50 * 1e3

    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit("bench.py: --scaling strong needs --batch divisible by --gpus")
        B = args.batch // world          # rows per GPU per step
    else:
        B = args.batch
    GB = B * world                       # global batch of one step
    theta, x = make_data(B, device, seed=rank)     # this rank's evaluation batch
    est = build_estimator(*make_data(BATCH, "cpu", seed=0), device)
    if distributed:
        broadcast_from_rank0(dist, est.net.flat_params.data)

    results = {}
    if args.mode in ("both", "log_prob"):
        def lp_step():
            with torch.no_grad():
                est.log_prob(theta, x)

        wall, dev_ms = timed(lp_step, args.steps, args.warmup, device, dist)
        results["log_prob"] = {"value": GB * args.steps / wall, "unit": "evals/s",
                               "ms_per_step": wall / args.steps * 1e3,
                               "roofline": roofline(F_EVAL, B, args.steps, dev_ms, "log_prob")}
    if args.mode in ("both", "log_prob_broadcast"):
        # M1 (ii): the samplers' shape -- N thetas against ONE observation (x is never expanded: 44 B / eval)
        x_o1 = x[:1].clone()
        th_s = theta.unsqueeze(1)        # (N, 1, D) sample-batch-event, condition (1, C)

        def lpb_step():
            with torch.no_grad():
                est.log_prob(th_s, x_o1)

        wall, dev_ms = timed(lpb_step, args.steps, args.warmup, device, dist)
        results["log_prob_broadcast_x"] = {"value": GB * args.steps / wall, "unit": "evals/s",
                                           "ms_per_step": wall / args.steps * 1e3,
                                           "roofline": roofline(F_EVAL, B, args.steps, dev_ms)}
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
        # M3 with acceptance < 1 (`BoxUniform(-1, 1)`; reference plumbing direct_posterior.py:177-213,
        # rejection.py:380-409): the support check rejects, accept_reject_sample loops with an adapted batch size and
        # compacts the accepted draws.  The estimator is the same random-init flow, so the boxes are chosen around what
        # IT puts out: the survey's (-1, 1) and a tight one for a low-acceptance regime; `acceptance` is measured.
        from sbi_amd.utils.sbiutils import within_support
        from sbi_amd.utils.torchutils import BoxUniform

        box_obj = {}
        for half in (1.0, 0.5):
            bprior = BoxUniform(-half * torch.ones(D, device=device), half * torch.ones(D, device=device))
            bpost = DirectPosterior(est, bprior, device=device)
            with torch.no_grad():
                acc = within_support(bprior, est.sample(torch.Size([nd]), condition=x_o)[:, 0]).float().mean().item()

            def box_step(bpost=bpost):
                bpost.sample((nd,), x=x_o, max_sampling_batch_size=nd, show_progress_bars=False)

            bsteps = max(1, min(args.steps, 5))
            wall_b, dev_b = timed(box_step, bsteps, 1, device, dist)
            box_obj[f"box_uniform_{half:g}"] = {
                "prior": f"BoxUniform(-{half:g}, {half:g})^{D}", "acceptance": acc, "value": nd * world * bsteps / wall_b,
                "unit": "accepted draws/s", "proposal_draws_per_s": nd * world * bsteps / wall_b / max(acc, 1e-9),
                "ms_per_step": wall_b / bsteps * 1e3, "device_ms_per_step": dev_b / bsteps, "steps": bsteps}
        results["sample"]["acceptance_below_one"] = box_obj
    if args.mode == "mcmc":
        out = mcmc_leg(est, x, device, with_cpu=(world == 1 and not args.no_cpu_baseline))
        if rank == 0:
            print(json.dumps(out))
        return
    if args.mode == "fmpe":
        out = fmpe_leg(args, B, rank, world, device, dist, distributed, GB)
        if rank == 0:
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = out.pop("_cpu_baseline_fn")()
            out.pop("_cpu_baseline_fn", None)
            print(json.dumps(out))
        if distributed:
            dist.destroy_process_group()
        return
    if args.mode in ("maf", "zuko"):
        # the maf_rqs sibling flow (MADE masked-linear conditioner, autoregressive RQ
        # splines) at the configs[1] shape: log_prob, sample for given noise (D conditioner passes per transform) and
        # the fused training step
        from sbi_amd.inference.trainers.fused import FusedTrainStep
        from sbi_amd.neural_nets.net_builders.flow import build_maf_rqs, build_zuko_nsf

        torch.manual_seed(1)
        th0, x0 = make_data(BATCH, "cpu", seed=0)
        flow_name = "maf_rqs" if args.mode == "maf" else "zuko_nsf"
        mest = (build_maf_rqs if args.mode == "maf" else build_zuko_nsf)(th0, x0).to(device)
        h = mest.net.hyper
        H, P_, NBm = h.hidden_features, 3 * h.num_bins - 1, h.num_blocks
        # dense FLOP per eval: 2 x (weights of the masked linears, masked-out entries included: the kernels run dense)
        f_eval = h.num_transforms * 2.0 * ((D + C) * H + NBm * H * H + H * D * P_)
        # inverse: D passes of [initial + blocks + ONE dim's final-layer rows] (maf_rqs: the context layer only once)
        f_draw = h.num_transforms * 2.0 * (C * H + D * (D * H + NBm * H * H + H * P_))
        noise = torch.randn(B, D, device=device)

        def m_lp():
            with torch.no_grad():
                mest.log_prob(theta, x)

        wall_lp, dev_lp = timed(m_lp, args.steps, args.warmup, device, dist)
        wall_s, dev_s = timed(lambda: mest.sample_from_noise(noise, x), max(1, args.steps // 5), 2, device, dist)
        stepper = FusedTrainStep(mest, lr=5e-4, clip_max_norm=5.0, distributed=distributed)
        wall_t, dev_t = timed(lambda: stepper.step(theta, x, global_batch=GB), args.steps, args.warmup, device, dist)
        if rank == 0:
            print(json.dumps({
                "metric": f"{flow_name} train (theta,x)-pairs/sec", "value": GB * args.steps / wall_t, "unit": "pairs/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall_t / args.steps * 1e3,
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": f"{flow_name} (sbi defaults: hidden {H}, {h.num_transforms} transforms, {NBm} "
                                       f"hidden-to-hidden layers, {h.num_bins} bins, {h.param_count()} parameters), "
                                       f"theta-dim {D}, "
                                       f"x-dim {C}, batch {B} per GPU", "parallelism": f"dp{world}"},
                "roofline": roofline(3 * f_eval, B, args.steps, dev_t),
                "log_prob": {"value": GB * args.steps / wall_lp, "unit": "evals/s",
                             "ms_per_step": wall_lp / args.steps * 1e3,
                             "roofline": roofline(f_eval, B, args.steps, dev_lp)},
                "sample_from_noise": {"value": GB * max(1, args.steps // 5) / wall_s, "unit": "draws/s",
                                      "ms_per_step": wall_s / max(1, args.steps // 5) * 1e3,
                                      "device_ms_per_step": dev_s / max(1, args.steps // 5),
                                      # no roofline object: pass i of the inverse only needs hidden units of
                                      # degree <= i, which the kernel exploits, so the dense count f_draw
                                      # (recorded for reference) over-states the work it does
                                      "dense_flop_per_draw": f_draw, "roofline": None}}))
        if distributed:
            dist.destroy_process_group()
        return
    if args.mode == "npe_train":
        out = npe_train_leg(device, rank, world, args.npe_epochs)
        if rank == 0:
            print(json.dumps(out))
        if distributed:
            dist.destroy_process_group()
        return
    if args.mode == "atomic":
        out = atomic_leg(est, theta, x, B, GB, args.steps, args.warmup, device, dist, distributed, world, args.scaling,
                         with_cpu=(world == 1 and not args.no_cpu_baseline))
        if rank == 0:
            print(json.dumps(out))
        if distributed:
            dist.destroy_process_group()
        return

    strong_obj = None
    if args.mode in ("both", "train"):
        # this rank's resident simulations: the 90 000-row training split of 100 000 simulations (weak scaling:
        # every rank its own 100 000; strong scaling: the ranks share ONE dataset and take 1/N of every batch)
        n_train = int(0.9 * N_SIMS)
        th_all, x_all = make_data(n_train, device, seed=1000 + (rank if args.scaling == "weak" else 0))
        leg = TrainLeg(est, th_all, x_all, B, distributed, GB)
        wall, dev_ms = timed(leg, args.steps, args.warmup, device, dist)
        # roofline: the fused step's kernels only (pack, forward, the backward launch over all T transforms, reduce, [all-reduce], clip+Adam),
        # HIP events on the launch stream around every step; the sampler's gather kernel is in `value` only
        fused_ms = leg.fused_ms(args.steps)
        results["train"] = {"value": GB * args.steps / wall, "unit": "pairs/s",
                            "ms_per_step": wall / args.steps * 1e3,
                            "roofline": roofline(F_TRAIN, B, args.steps, fused_ms, "train")}
        results["train"]["roofline"]["whole_step_device_ms"] = dev_ms / args.steps
        results["train"]["host"] = dict(LAST_TIMED)
        if world > 1 and args.scaling == "weak" and args.batch % world == 0:
            # the same inner loop with the 65 536-pair global batch split over the ranks
            Bs = args.batch // world
            leg_s = TrainLeg(est, th_all, x_all, Bs, distributed, args.batch)
            wall_s, dev_ms_s = timed(leg_s, args.steps, args.warmup, device, dist)
            strong_obj = {"scaling": "strong", "baseline_config": "BASELINE configs[2]: the 65 536-pair batch of "
                          "configs[1] (100 000 simulations cannot feed more than one such batch) split over the GPUs",
                          "global_batch": args.batch, "rows_per_gpu": Bs,
                          "value": args.batch * args.steps / wall_s, "unit": "pairs/s",
                          "ms_per_step": wall_s / args.steps * 1e3,
                          "roofline": roofline(F_TRAIN, Bs, args.steps, leg_s.fused_ms(args.steps))}

    small_obj = None
    if args.mode in ("both", "train") and world == 1 and not distributed and not args.no_small_batch:
        # The latency regime (cooperative kernels, csrc/nsf_coop.h): sbi's default training_batch_size = 200
        # (npe_base.py:301-316) and the per-GPU share of SURVEY 8(e)'s partitioning -- the 65 536-pair batch split over
        # 8 GPUs = 8 192 pairs per GPU per step.  Same inner loop as the headline leg (shuffled gather, fused step).
        small_obj = {}
        for b in (200, 8192):
            leg_b = TrainLeg(est, th_all, x_all, b, False, b)
            ks = 200
            wall_b, _ = timed(leg_b, ks, 20, device)
            small_obj[f"batch_{b}"] = {"ms_per_step": wall_b / ks * 1e3, "fused_step_device_ms": leg_b.fused_ms(ks) / ks,
                                       "value": b * ks / wall_b, "unit": "pairs/s"}
        small_obj["strong_scaling_8_gpus"] = {
            "rows_per_gpu": 8192, "fused_step_ms_at_65536_rows": results["train"]["roofline"]["device_ms_per_step"],
            "fused_step_ms_at_8192_rows": small_obj["batch_8192"]["fused_step_device_ms"],
            "speedup_bound_before_allreduce": results["train"]["roofline"]["device_ms_per_step"]
            / small_obj["batch_8192"]["fused_step_device_ms"],
            "note": "one 65 536-pair step on 1 GPU vs the same step's 8 192-pair share on each of 8 GPUs; the RCCL "
                    "all-reduce of the 392 KB gradient comes on top (rccl_1rank.allreduce_us_98025_floats)"}
        # hidden_features = 100 (wide cooperative kernels, csrc/nsf_coop_wide_kernel.h): same inner loop, same data
        from sbi_amd.neural_nets.net_builders.flow import build_nsf

        torch.manual_seed(0)
        est_w = build_nsf(th_all[:4096].cpu(), x_all[:4096].cpu(), hidden_features=100).to(device)
        wide_obj = {"hidden_features": 100, "parameters": int(est_w.net.flat_params.numel())}
        for b in (200, 8192):
            leg_w = TrainLeg(est_w, th_all, x_all, b, False, b)
            ks = 100
            wall_w, _ = timed(leg_w, ks, 10, device)
            tb, xb = th_all[:b].contiguous(), x_all[:b].contiguous()
            _, lp_ms = timed(lambda: est_w.log_prob(tb, xb), ks, 10, device)
            wide_obj[f"batch_{b}"] = {"train_ms_per_step": wall_w / ks * 1e3, "fused_step_device_ms": leg_w.fused_ms(ks) / ks,
                                      "train_value": b * ks / wall_w, "unit": "pairs/s", "log_prob_device_ms": lp_ms / ks}
        small_obj["wide_hidden_100"] = wide_obj

    atomic_obj = mcmc_obj = None
    if args.mode == "both" and world == 1 and not distributed:
        # SURVEY 8(f) rows 2 and 3 where the driver sees them: short legs, nested like the other secondary metrics
        atomic_obj = atomic_leg(est, theta[:8192].contiguous(), x[:8192].contiguous(), 8192, 8192, min(args.steps, 20),
                                min(args.warmup, 5), device, None, False, 1, "weak", with_cpu=not args.no_cpu_baseline)
        mcmc_obj = mcmc_leg(est, x, device, with_cpu=not args.no_cpu_baseline)
    npe_obj = npe_train_leg(device, rank, world, args.npe_epochs) if args.mode == "both" else None
    fm_out = fmpe_leg(args, B, rank, world, device, dist, distributed, GB) if args.mode == "both" else None
    if rank == 0:
        head = next(k for k in ("train", "log_prob", "log_prob_broadcast_x", "sample") if k in results)
        r = results[head]
        per = "per GPU" if args.scaling == "weak" else f"global, {B} per GPU"
        out = {
            "metric": {"train": "NPE train (theta,x)-pairs/sec", "log_prob": "NSF log_prob evals/sec",
                       "log_prob_broadcast_x": "NSF log_prob evals/sec (one x_o for all rows)",
                       "sample": "DirectPosterior.sample draws/sec"}[head],
            "value": r["value"], "unit": r["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # untimed steps of the same leg run BEFORE the W warm-up steps so that the K timed steps do not sit in the
            # first ~40 ms of a device that was idle (see preheat()); every leg's timed() does the same
            "preheat": {"target_ms": PREHEAT_MS, "untimed_steps_before_warmup": r.get("host", {}).get("preheat_steps", LAST_TIMED.get("preheat_steps"))},
            "config": {"workload": f"BASELINE configs[1]: NPE + NSF theta-dim {D}, x-dim {C}, {N_SIMS} simulations "
                                   f"(90 000-row training split resident in HBM), batch {args.batch} {per}, synthetic "
                                   f"linear-Gaussian; step = shuffled batch gather (device sampler) + fused NPE training "
                                   f"step (pack, loss fwd+bwd, grad all-reduce, clip+Adam)"
                       if head == "train" else
                       f"NSF log_prob, theta-dim {D}, x-dim {C}, batch {args.batch} {per}",
                       "parallelism": f"dp{world}", "rccl_ranks": rccl_world if distributed else 1,
                       "rccl_allreduce_us_98025_floats": allreduce_us,
                       "baseline_config": ("configs[1]" if world == 1 else
                                           "configs[2] (strong: one 65 536-pair batch split over the GPUs)"
                                           if args.scaling == "strong" else
                                           "configs[1] replicated per GPU (weak); configs[2] itself is the nested "
                                           "`strong_scaling` object"),
                       **({"shared_gpu": "all ranks on device 0 over gloo (SBI_AMD_BENCH_SHARE_GPU=1): a code-path "
                                         "run, not a scaling measurement"} if share_gpu else {})},
            # whole step (forward + one backward launch over the T transforms + reduce + clip/Adam) against dense fp32 MFMA;
            # per-kernel durations: profiles/*kernel_stats.csv
            "roofline": r["roofline"],
        }
        if "host" in r:
            out["host"] = r["host"]
        if strong_obj is not None:
            out["strong_scaling"] = strong_obj
        for key in ("log_prob", "log_prob_broadcast_x"):
            if key in results and head == "train":
                lp = results[key]
                out[key] = {"metric": "NSF log_prob evals/sec" + (" (one x_o for all rows)" if "broadcast" in key
                                                                  else " (paired x)"),
                            "value": lp["value"], "unit": lp["unit"], "ms_per_step": lp["ms_per_step"],
                            "roofline": lp["roofline"]}
        if "sample" in results and head != "sample":
            sp = results["sample"]
            out["posterior_sample"] = {"metric": "DirectPosterior.sample draws/sec", "value": sp["value"],
                                       "unit": sp["unit"], "draws_per_step": args.draws, "steps": sp["steps"],
                                       "ms_per_step": sp["ms_per_step"], "roofline": sp["roofline"],
                                       "prior": "Gaussian N(0, 0.1 I): acceptance 1 (support check still runs)",
                                       "acceptance_below_one": sp.get("acceptance_below_one")}
        if small_obj is not None:
            out["small_batch"] = small_obj
        if world == 1 and not distributed and head == "train" and not args.no_rccl_leg:
            out["rccl_1rank"] = rccl_one_rank_leg(args, r)
        if npe_obj is not None:
            out["npe_train"] = npe_obj
        if atomic_obj is not None:
            out["npe_c_atomic"] = {k: atomic_obj[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "roofline",
                                                              "cpu_baseline") if k in atomic_obj}
            out["npe_c_atomic"]["workload"] = atomic_obj["config"]["workload"]
        if mcmc_obj is not None:
            out["mcmc"] = {k: v for k, v in mcmc_obj.items() if k not in ("n_gpus", "config")}
            out["mcmc"]["workload"] = mcmc_obj["config"]["workload"]
        if fm_out is not None:
            out["fmpe_train"] = {k: fm_out[k] for k in ("metric", "value", "unit", "ms_per_step", "roofline",
                                                        "host", "posterior_sample")}
            out["fmpe_train"]["workload"] = fm_out["config"]["workload"]
            if world == 1 and not args.no_cpu_baseline:
                out["fmpe_train"]["cpu_baseline"] = fm_out["_cpu_baseline_fn"]()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(head)
            if head == "train":
                # `cpu_baseline` above feeds the oracle PRE-BATCHED tensors (the generous baseline); the reference's own
                # loop collates every batch element by element -- timed next to it, both labelled
                out["cpu_baseline"]["loader"] = "pre-batched tensors (no DataLoader): the generous variant"
                out["cpu_baseline_reference_loader"] = cpu_baseline("train_dataloader", budget_s=8.0)
            if "posterior_sample" in out:
                out["posterior_sample"]["cpu_baseline"] = cpu_baseline("sample", budget_s=6.0)
            if "log_prob" in results and head == "train":
                out["log_prob"]["cpu_baseline"] = cpu_baseline("log_prob", budget_s=8.0)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
