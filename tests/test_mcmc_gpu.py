"""Device-resident vectorised slice sampler and MCMCPosterior (reference:
sbi/samplers/mcmc/slice_numpy.py:353-587, sbi/inference/posteriors/mcmc_posterior.py, tests/mcmc_test.py)."""

import warnings

import pytest
import torch
from torch.distributions import MultivariateNormal

from sbi_amd.inference import NPE
from sbi_amd.neural_nets import NSFConfig
from sbi_amd.samplers.mcmc import SliceSamplerVectorized
from sbi_amd.simulators.linear_gaussian import linear_gaussian, true_posterior_linear_gaussian_mvn_prior
from sbi_amd.utils.metrics import c2st
from sbi_amd.utils.torchutils import BoxUniform

pytestmark = pytest.mark.gpu


def test_slice_sampler_recovers_a_correlated_gaussian():
    """tests/mcmc_test.py:41-80 (`test_c2st_slice_np_on_Gaussian`): target N(mean, cov), c2st vs exact draws."""
    torch.manual_seed(0)
    dim, chains, per_chain = 3, 64, 60
    mean = torch.tensor([1.0, -2.0, 0.5], device="cuda")
    a = torch.tensor([[1.0, 0.6, 0.0], [0.0, 1.0, -0.4], [0.0, 0.0, 0.7]], device="cuda")
    cov = a @ a.T
    target = MultivariateNormal(mean, cov)
    sampler = SliceSamplerVectorized(log_prob_fn=target.log_prob, init_params=torch.randn(chains, dim, device="cuda"),
                                     num_chains=chains, thin=3, tuning=50)
    samples = sampler.run(per_chain * 3 + 60)[:, 20:, :]          # thinned; drop the first 20 kept sweeps
    assert samples.shape == (chains, per_chain, dim)
    flat = samples.reshape(-1, dim)
    assert torch.allclose(flat.mean(0), mean, atol=0.12)
    assert torch.allclose(torch.cov(flat.T), cov, atol=0.2)
    score = c2st(flat[:2000].cpu(), target.sample((2000,)).cpu()).item()
    print(f"slice sampler c2st={score:.3f} ticks={sampler.num_ticks}")
    assert 0.4 <= score <= 0.6
    # bracket widths were tuned away from the 0.01 start
    assert (sampler.width > 0.1).all()


def test_slice_sampler_respects_minus_inf_regions():
    """A potential that is -inf outside a box never yields samples outside of it (support handling of the
    posterior-based potential)."""
    torch.manual_seed(1)
    lo, hi = -1.0, 2.0

    def log_prob(th):
        inside = ((th > lo) & (th < hi)).all(dim=1)
        return torch.where(inside, -0.5 * (th**2).sum(1), torch.full_like(th[:, 0], float("-inf")))

    sampler = SliceSamplerVectorized(log_prob, torch.zeros(32, 2, device="cuda"), num_chains=32, tuning=20)
    s = sampler.run(100).reshape(-1, 2)
    assert (s > lo).all() and (s < hi).all() and torch.isfinite(s).all()


@pytest.mark.parametrize("prior_kind", ["gaussian", "uniform"])
def test_mcmc_posterior_matches_direct_posterior(prior_kind):
    """NPE + NSF trained once; `sample_with="mcmc"` (unconstrained-space slice sampling of the estimator's
    log-prob) against the analytic posterior (tests/linearGaussian_snpe_test.py:501-562)."""
    dim, n = 2, 2500
    torch.manual_seed(0)
    shift, cov = -1.0 * torch.ones(dim), 0.3 * torch.eye(dim)
    if prior_kind == "gaussian":
        prior = MultivariateNormal(torch.zeros(dim, device="cuda"), torch.eye(dim, device="cuda"))
    else:
        prior = BoxUniform(-2.0 * torch.ones(dim), 2.0 * torch.ones(dim), device="cuda")
    theta = prior.sample((n,)).cpu()
    x = linear_gaussian(theta, shift, cov)
    x_o = torch.zeros(1, dim)
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator=NSFConfig(), device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=100)
    direct = inf.build_posterior().set_default_x(x_o)
    mcmc = inf.build_posterior(sample_with="mcmc", mcmc_method="slice_np_vectorized",
                               mcmc_parameters=dict(num_chains=100, thin=2, warmup_steps=20,
                                                    init_strategy="resample")).set_default_x(x_o)
    s_mcmc = mcmc.sample((1000,), show_progress_bars=False)
    assert s_mcmc.shape == (1000, dim) and bool(prior.support.check(s_mcmc).all())
    s_direct = direct.sample((1000,), show_progress_bars=False)
    score = c2st(s_mcmc.cpu(), s_direct.cpu()).item()
    print(f"{prior_kind}: c2st(mcmc, direct)={score:.3f}")
    assert 0.4 <= score <= 0.6
    if prior_kind == "gaussian":
        target = true_posterior_linear_gaussian_mvn_prior(x_o, shift, cov, torch.zeros(dim), torch.eye(dim)).sample((1000,))
        assert 0.4 <= c2st(s_mcmc.cpu(), target).item() <= 0.62
    # chains can be continued, the potential is the unnormalised log-density
    again = mcmc.sample((200,), init_strategy="latest_sample", num_chains=50, show_progress_bars=False)
    assert again.shape == (200, dim)
    pot = mcmc.potential(s_direct[:10])
    assert torch.allclose(pot, direct.log_prob(s_direct[:10], norm_posterior=False), atol=1e-4)
    with pytest.raises(NotImplementedError):
        mcmc.sample((10,), method="nuts_pyro")


def _torch_tick(st, logp, u, num_samples, tuning, max_width):
    """Tensorised restatement of the per-chain transitions of slice_numpy.py:438-566 (test oracle for the
    tick kernel; consumes the same uniforms)."""
    x, nxt, width, order, state, i, t, cxi, wi, lx, ux, xi, logu, samples = (st[k] for k in (
        "x", "nxt", "width", "order", "state", "i", "t", "cxi", "wi", "lx", "ux", "xi", "logu", "samples"))
    C, D = x.shape
    ar = torch.arange(C, device=x.device)
    dim = order[ar, i]
    live = state != 4
    is_b, is_l, is_u, is_s = (live & (state == k) for k in range(4))
    # BEGIN
    cxi = torch.where(is_b, x[ar, dim], cxi)
    wi = torch.where(is_b, width[ar, dim], wi)
    logu = torch.where(is_b, logp + torch.log(1.0 - u[:, 0]), logu)
    lx_b = cxi - wi * u[:, 1]
    # LOWER
    out_l = is_l & (logp >= logu) & (cxi - lx < max_width)
    # UPPER
    out_u = is_u & (logp >= logu) & (ux - cxi < max_width)
    # SAMPLE
    rej = is_s & (logp < logu)
    acc = is_s & ~rej
    new_lx = torch.where(is_b, lx_b, torch.where(out_l, lx - wi, torch.where(rej & (xi < cxi), xi, lx)))
    new_ux = torch.where(is_b, lx_b + wi, torch.where(out_u, ux + wi, torch.where(rej & ~(xi < cxi), xi, ux)))
    draw = (new_ux - new_lx) * u[:, 2] + new_lx
    new_xi = torch.where((is_u & ~out_u) | rej, draw, xi)
    val = torch.where(is_b | out_l, new_lx, torch.where((is_l & ~out_l) | out_u, new_ux, new_xi))
    write = live & ~acc
    nxt[ar[write], dim[write]] = val[write]
    x[ar[acc], dim[acc]] = xi[acc]
    tune = acc & (t < tuning)
    w_old = width[ar[tune], dim[tune]]
    width[ar[tune], dim[tune]] = w_old + ((ux[tune] - lx[tune]) - w_old) / (t[tune] + 1).float()
    sweep_end = acc & (i == D - 1)
    store = sweep_end & (t >= tuning)
    samples[ar[store], (t[store] - tuning)] = x[store]
    # fresh order by Fisher-Yates on the same uniforms
    for c in ar[sweep_end].tolist():
        o = list(range(D))
        for d in range(D - 1, 0, -1):
            k = min(int(float(u[c, 4 + d]) * (d + 1)), d)        # float32 product as in the kernel
            o[d], o[k] = o[k], o[d]
        order[c] = torch.tensor(o, dtype=order.dtype, device=order.device)
    new_state = torch.where(is_b, 1, torch.where(is_l & ~out_l, 2, torch.where(is_u & ~out_u, 3,
                            torch.where(acc, 0, state))))
    t = torch.where(sweep_end, t + 1, t)
    i = torch.where(acc, torch.where(sweep_end, torch.zeros_like(i), i + 1), i)
    new_state = torch.where(sweep_end & (t >= num_samples + tuning), 4, new_state)
    st.update(state=new_state, i=i, t=t, cxi=cxi, wi=wi, lx=new_lx, ux=new_ux, xi=new_xi, logu=logu)


def test_tick_kernel_matches_tensorised_restatement_bit_for_bit():
    from sbi_amd import _lib

    torch.manual_seed(3)
    lib = _lib.load()
    C, D, NS, TUNE, MAXW = 37, 4, 6, 3, 3.0e38
    dev = "cuda"
    f = lambda th: -0.5 * ((th - 0.3) ** 2 / torch.tensor([1.0, 0.25, 4.0, 0.5], device=dev)).sum(1)
    x0 = torch.randn(C, D, device=dev)
    order0 = torch.rand(C, D, device=dev).argsort(1).to(torch.int32)
    # kernel state
    x, nxt = x0.clone(), x0.clone()
    width = torch.full((C, D), 0.7, device=dev)
    order = order0.clone()
    istate = torch.zeros(C, 4, dtype=torch.int32, device=dev)
    fstate = torch.zeros(C, 8, device=dev)
    samples = torch.zeros(C, NS, D, device=dev)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    # restatement state
    st = dict(x=x0.clone(), nxt=x0.clone(), width=torch.full((C, D), 0.7, device=dev), order=order0.clone().long(),
              state=torch.zeros(C, dtype=torch.long, device=dev), i=torch.zeros(C, dtype=torch.long, device=dev),
              t=torch.zeros(C, dtype=torch.long, device=dev), samples=torch.zeros(C, NS, D, device=dev),
              **{k: torch.zeros(C, device=dev) for k in ("cxi", "wi", "lx", "ux", "xi", "logu")})
    for tick in range(4000):
        u = torch.rand(C, 4 + D, device=dev)
        logp = f(nxt).contiguous()
        assert torch.equal(logp, f(st["nxt"]))
        rc = lib.sbi_amd_mcmc_slice_tick(C, D, NS, TUNE, MAXW, _lib.ptr(logp), None, _lib.ptr(u), _lib.ptr(x), _lib.ptr(nxt),
                                         _lib.ptr(width), _lib.ptr(order), _lib.ptr(istate), _lib.ptr(fstate),
                                         _lib.ptr(samples), _lib.ptr(done), 0, 0, 0, None, None, None, None, _lib.current_stream(torch.device(dev)))
        assert rc == 0
        _torch_tick(st, logp, u, NS, TUNE, MAXW)
        assert torch.equal(istate[:, 0].long(), st["state"]), tick
        assert torch.equal(nxt, st["nxt"]) and torch.equal(x, st["x"]), tick
        if int(done.item()) == C:
            break
    assert int(done.item()) == C and bool((st["state"] == 4).all())
    assert torch.equal(samples, st["samples"]) and torch.equal(width, st["width"])
    assert torch.equal(order.long(), st["order"])


def test_persistent_sampler_equals_the_two_launch_loop():
    """sbi_amd_mcmc_slice_run (one launch per `poll_every` ticks, a workgroup owns 16 chains) against the loop of
    log_prob + tick launches: same Philox counters, same log-density kernel -> the same chains, bit for bit."""
    from torch.distributions import MultivariateNormal

    from sbi_amd.inference.posteriors.mcmc_posterior import MCMCPosterior
    from sbi_amd.inference.potentials.posterior_based_potential import posterior_estimator_based_potential
    from sbi_amd.samplers.mcmc import SliceSamplerVectorized
    from sbi_amd.utils.sbiutils import mcmc_transform
    from tests.helpers import matched_pair

    _, est, _, x = matched_pair(D=4, C=3)
    prior = MultivariateNormal(torch.zeros(4, device="cuda"), torch.eye(4, device="cuda"))
    potential_fn, _ = posterior_estimator_based_potential(est, prior, x_o=None)
    tf = mcmc_transform(prior, device="cuda")
    post = MCMCPosterior(potential_fn, prior, tf, num_chains=100, thin=1, warmup_steps=3, device="cuda")
    x_o = x[:1].cuda()
    post.set_default_x(x_o)
    post.potential_fn.set_x(x_o, x_is_iid=True)
    fused = post._fused_potential()
    assert fused is not None and len(fused.fused_spec) == 6
    init = torch.randn(100, 4, device="cuda") * 0.3
    outs = []
    for persistent in (True, False):
        torch.manual_seed(11)
        s = SliceSamplerVectorized(fused, init.clone(), num_chains=100, thin=1, tuning=10, poll_every=16)
        s.persistent = persistent
        outs.append((s.run(12).clone(), s.num_ticks))
    assert outs[0][1] >= outs[1][1] and outs[0][1] - outs[1][1] < 16
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.isfinite(outs[0][0]).all() and outs[0][0].std() > 0.05
