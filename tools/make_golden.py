#!/usr/bin/env python
"""Generate tests/golden/*.pt from the REAL reference code (run in the build container only).

The reference's NSF arithmetic lives in nflows (not installed, not vendored), but the
pieces of the hot path that live in the reference TREE can be imported once the
missing third-party imports are stubbed: z-score statistics, masks, searchsorted,
shape handling, the linear-Gaussian simulator / analytic posterior, within_support.
This script imports them from /root/reference, evaluates them on seeded inputs and
stores inputs + outputs; tests/test_golden_reference.py replays them against
sbi_amd and the oracle.  /root/reference is NOT needed at test time.
"""

import importlib.machinery
import os
import sys
import types

import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _Anything:
    """Class-like placeholder: subclassable, callable, attribute chain never fails."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if full in sys.modules:
            return sys.modules[full]
        return type(name, (), {"__init__": lambda self, *a, **k: None})


def stub(name):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        n = ".".join(parts[:i])
        if n not in sys.modules:
            m = _StubModule(n)
            m.__path__ = []
            m.__spec__ = importlib.machinery.ModuleSpec(n, None, is_package=True)
            sys.modules[n] = m


for mod in ["nflows", "nflows.transforms", "nflows.flows", "nflows.nn", "nflows.nn.nets", "nflows.distributions",
            "nflows.utils", "nflows.nn.nde", "nflows.nn.nde.made", "nflows.transforms.splines",
            "nflows.transforms.splines.rational_quadratic", "nflows.transforms.base",
            "zuko", "zuko.flows", "zuko.transforms", "zuko.distributions", "zuko.lazy", "zuko.flows.core",
            "zuko.flows.autoregressive", "zuko.flows.spline", "zuko.nn", "zuko.utils", "zuko.mixtures",
            "zuko.flows.mixture",
            "pyro", "pyro.distributions", "pyro.distributions.transforms", "pyro.infer", "pyro.infer.mcmc",
            "pyro.infer.mcmc.api", "pyro.distributions.empirical", "pyknos", "pyknos.nflows",
            "pyknos.mdn", "pyknos.mdn.mdn", "pyknos.nflows.transforms", "pyknos.nflows.nn",
            "torch.utils.tensorboard", "torch.utils.tensorboard.writer", "tensorboard", "arviz", "pymc",
            "skorch", "tabpfn", "tabpfn_extensions"]:
    stub(mod)
sys.path.insert(0, REF)


def main():
    os.makedirs(OUT, exist_ok=True)
    g = {}
    from sbi.utils import torchutils as T   # noqa: E402

    torch.manual_seed(0)
    # searchsorted incl. the reference test's own vectors (tests/torchutils_test.py:138-158)
    bins = torch.linspace(0, 1, 10)
    left, right = bins[:-1], bins[1:]
    mid = left + 0.5 * (right - left)
    g["searchsorted"] = dict(bins=bins.clone(), left=left.clone(), right=right.clone(), mid=mid.clone(),
                             idx_left=T.searchsorted(bins[None, :].clone(), left),
                             idx_right=T.searchsorted(bins[None, :].clone(), right),
                             idx_mid=T.searchsorted(bins[None, :].clone(), mid))
    knots = torch.sort(torch.rand(64, 5, 11) * 6 - 3, dim=-1).values
    xs = torch.rand(64, 5) * 6 - 3
    g["searchsorted_rand"] = dict(knots=knots.clone(), x=xs.clone(), idx=T.searchsorted(knots.clone(), xs))
    g["masks"] = {f"{d}_{int(e)}": T.create_alternating_binary_mask(d, even=e) for d in (2, 3, 4, 7, 10)
                  for e in (True, False)}
    x = torch.randn(3, 4)
    g["repeat_rows"] = dict(x=x, out=T.repeat_rows(x, 3))
    g["sum_except_batch"] = dict(x=torch.randn(5, 3, 2), )
    g["sum_except_batch"]["out"] = T.sum_except_batch(g["sum_except_batch"]["x"])
    box = T.BoxUniform(-2 * torch.ones(3), 2 * torch.ones(3))
    pts = torch.randn(50, 3) * 2
    from sbi.utils import sbiutils as S   # noqa: E402

    g["within_support"] = dict(pts=pts, inside=S.within_support(box, pts), logp=box.log_prob(pts))

    # z-score statistics (theta side: z_standardization; x side: standardizing_net buffers)
    batch = torch.randn(257, 6) * torch.tensor([1.0, 10.0, 1e-3, 5.0, 0.0, 2.0]) + torch.tensor(
        [0.0, 3.0, -1.0, 100.0, 7.0, 0.5])
    batch[5, 2] = float("nan")
    batch[9, 0] = float("inf")
    z = {}
    for structured in (False, True):
        m, s = S.z_standardization(batch, structured)
        net = S.standardizing_net(batch, structured)
        z[f"theta_{structured}"] = dict(mean=m, std=s)
        z[f"x_{structured}"] = dict(mean=net._mean.clone(), std=net._std.clone())
    one = S.standardizing_net(batch[:1])
    z["x_single_row"] = dict(mean=one._mean.clone(), std=one._std.clone())
    g["zscore"] = dict(batch=batch, stats=z)
    g["z_score_parser"] = {str(k): S.z_score_parser(k) for k in (None, "none", "independent", "structured",
                                                                  "transform_to_unconstrained")}
    hv = torch.randn(20, 3)
    hv[3, 1] = float("nan")
    hv[7, 0] = float("-inf")
    g["handle_invalid_x"] = dict(x=hv, out=S.handle_invalid_x(hv, True))

    # shape handling
    from sbi.neural_nets.estimators import shape_handling as H   # noqa: E402

    sh = {}
    for name, t, ev, lead in [("e", torch.randn(4), (4,), False), ("be", torch.randn(3, 4), (4,), False),
                              ("se", torch.randn(3, 4), (4,), True), ("sbe", torch.randn(2, 3, 4), (4,), False),
                              ("sbe_l", torch.randn(2, 3, 4), (4,), True)]:
        sh[name] = dict(inp=t, out=H.reshape_to_sample_batch_event(t, torch.Size(ev), leading_is_sample=lead),
                        lead=lead)
    g["shape_handling"] = sh

    # linear Gaussian simulator + analytic posterior (tests/linearGaussian_snpe_test.py:60-92 setup)
    import importlib

    LG = importlib.import_module("sbi.simulators.linear_gaussian")

    lg = {}
    for dim in (2, 10):
        shift = -1.0 * torch.ones(dim)
        cov = 0.3 * torch.eye(dim)
        x_o = torch.zeros(1, dim)
        post = LG.true_posterior_linear_gaussian_mvn_prior(x_o, shift, cov, torch.zeros(dim), torch.eye(dim))
        torch.manual_seed(7)
        theta = torch.randn(16, dim)
        sim = LG.linear_gaussian(theta, shift, cov)
        lg[dim] = dict(mean=post.mean, cov=post.covariance_matrix, theta=theta, sim=sim)
    post = LG.true_posterior_linear_gaussian_mvn_prior(torch.full((1, 10), 0.25), torch.zeros(10),
                                                       0.1 * torch.eye(10), torch.zeros(10), 0.1 * torch.eye(10))
    lg["mini_sbibm"] = dict(mean=post.mean, cov=post.covariance_matrix)
    g["linear_gaussian"] = lg

    # the real accept_reject_sample on a seeded toy proposal (rejection.py:230-457)
    R = importlib.import_module("sbi.samplers.rejection.rejection")

    def proposal(shape, condition):
        return torch.randn(shape[0], condition.shape[0], 3) * 0.8

    cond = torch.zeros(2, 5)
    torch.manual_seed(3)
    smp, acc = R.accept_reject_sample(proposal, lambda t: S.within_support(box, t), 5000,
                                      max_sampling_batch_size=700, proposal_sampling_kwargs={"condition": cond})
    g["accept_reject"] = dict(samples=smp, acceptance=acc)
    torch.save(g, os.path.join(OUT, "reference_intree.pt"))
    print("wrote", os.path.join(OUT, "reference_intree.pt"), {k: type(v).__name__ for k, v in g.items()})


if __name__ == "__main__":
    main()
