"""Replay of the REAL reference's `SliceSamplerVectorized` and `mcmc_transform` (fixtures written by
tools/make_golden_mcmc.py, which imports them from the reference tree): the HIP tick kernel fed the same
uniforms must walk the same trajectories."""

import os

import pytest
import torch

from sbi_amd.utils.sbiutils import mcmc_transform
from sbi_amd.utils.torchutils import BoxUniform

G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "mcmc_reference.pt"), weights_only=False)


def _priors():
    mvn = torch.distributions.MultivariateNormal(torch.tensor([1.0, -1.0, 0.5]), torch.diag(torch.tensor([4.0, 0.25, 1.0])))
    box = BoxUniform(-2.0 * torch.ones(3), torch.tensor([3.0, 1.0, 0.0]))
    return dict(mvn=mvn, box=box)


def test_mcmc_transform_matches_reference_values():
    for name, prior in _priors().items():
        ref = G["transform"][name]
        tf = mcmc_transform(prior)
        u = tf(ref["theta"])
        assert torch.allclose(u, ref["u"], atol=1e-5, rtol=1e-5), name
        assert torch.allclose(tf.log_abs_det_jacobian(ref["theta"], u), ref["lad"], atol=1e-5, rtol=1e-5), name
        assert torch.allclose(tf.inv(ref["u"]), ref["theta"], atol=1e-5, rtol=1e-5), name


@pytest.mark.gpu
def test_to_constrained_kernel_matches_reference_transform():
    from sbi_amd import _lib

    lib = _lib.load()
    for name, kind in (("mvn", 1), ("box", 2)):
        ref = G["transform"][name]
        prior = _priors()[name]
        if kind == 1:
            p0, p1 = prior.mean.cuda().contiguous(), prior.stddev.cuda().contiguous()
        else:
            low, high = prior.base_dist.low, prior.base_dist.high
            p0, p1 = low.cuda().contiguous(), (high - low).cuda().contiguous()
        u = ref["u"].cuda().contiguous()
        theta = torch.empty_like(u)
        lad = torch.empty(u.shape[0], device="cuda")
        rc = lib.sbi_amd_mcmc_to_constrained(kind, u.shape[0], u.shape[1], _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(u),
                                             _lib.ptr(theta), _lib.ptr(lad), _lib.current_stream(u.device))
        assert rc == 0
        assert torch.allclose(theta.cpu(), ref["theta"], atol=2e-5, rtol=1e-5), name
        assert torch.allclose(lad.cpu(), ref["lad"], atol=2e-5, rtol=1e-5), name


@pytest.mark.gpu
def test_tick_kernel_walks_the_reference_sampler_trajectories():
    from sbi_amd import _lib

    lib = _lib.load()
    C, D, NS, TUNE = G["C"], G["D"], G["num_samples"], G["tuning"]
    dev = torch.device("cuda")
    w, center = G["weights"].to(dev), G["center"]
    f = lambda th: -0.5 * ((th - center) ** 2 / w).sum(1)
    x = G["init"].to(dev).contiguous()
    nxt = x.clone()
    width = torch.full((C, D), G["init_width"], device=dev)
    order = G["order0"].to(dev).to(torch.int32).contiguous()
    istate = torch.zeros(C, 4, dtype=torch.int32, device=dev)
    fstate = torch.zeros(C, 8, device=dev)
    samples = torch.zeros(C, NS, D, device=dev)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    table = G["table"].to(dev)
    for tick in range(table.shape[0]):
        logp = f(nxt).contiguous()
        u = table[tick].contiguous()
        rc = lib.sbi_amd_mcmc_slice_tick(C, D, NS, TUNE, 3.0e38, _lib.ptr(logp), None, _lib.ptr(u), _lib.ptr(x),
                                         _lib.ptr(nxt), _lib.ptr(width), _lib.ptr(order), _lib.ptr(istate),
                                         _lib.ptr(fstate), _lib.ptr(samples), _lib.ptr(done), 0, 0, 0, None, None, None, None, _lib.current_stream(dev))
        assert rc == 0
        if tick % 32 == 31 and int(done.item()) == C:
            break
    assert int(done.item()) == C
    # float64 numpy in the reference vs float32 here: same accept/reject decisions, values to ~1e-6
    assert torch.allclose(samples.cpu().double(), G["samples"], atol=1e-4, rtol=1e-4)
    assert torch.allclose(width.cpu().double(), G["widths"], atol=1e-4, rtol=1e-4)
