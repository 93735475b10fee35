"""Host-side behaviour (no GPU): estimator contract, builders/configs, trainer loop
semantics, rejection sampler, DirectPosterior -- modelled on the reference's
tests/density_estimator_test.py, factory_config_test.py, base_test.py,
rejection_sampling_test.py and linearGaussian_snpe_test.py plumbing."""

import pickle
import warnings

import pytest
import torch

from sbi_amd.inference import NPE, DirectPosterior
from sbi_amd.neural_nets import NSFConfig, posterior_nn
from sbi_amd.neural_nets.net_builders.flow import build_nsf
from sbi_amd.utils.torchutils import BoxUniform
from tests.helpers import linear_gaussian_data
from tests.oracle_adapter import OracleEstimator, oracle_build_fn


# ---------------------------------------------------------------- contract / shapes
def test_nsf_flow_rejects_cpu_tensors_loudly():
    theta, x = linear_gaussian_data(50, 4, 7)
    est = build_nsf(theta, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        est.log_prob(theta[:4], x[:4])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        est.sample((3,), x[:1])


def test_shape_errors_follow_reference_messages():
    theta, x = linear_gaussian_data(50, 4, 7)
    est = build_nsf(theta, x)
    with pytest.raises(ValueError, match="does not match the expected input dimensionality"):
        est.log_prob(theta[:4, :3], x[:4])
    with pytest.raises(ValueError, match="Shape of condition"):
        est.log_prob(theta[:4], x[:4, :5])
    with pytest.raises(RuntimeError, match="broadcastable batch dimensions"):
        est.log_prob(theta[:4], x[:3])
    assert est.input_shape == torch.Size([4]) and est.condition_shape == torch.Size([7])


def test_broadcast_dims_matrix():
    theta, x = linear_gaussian_data(50, 4, 7)
    est = build_nsf(theta, x)
    th, xx, S, B = est._flatten_pair(theta[:12].reshape(3, 4, 4), x[:4])
    assert (S, B) == (3, 4) and th.shape == (12, 4) and xx.shape == (4, 7)
    th, xx, S, B = est._flatten_pair(theta[:12].unsqueeze(1), x[:1])
    assert (S, B) == (12, 1) and xx.shape == (1, 7)          # single x_o is NOT materialised 12 times
    th, xx, S, B = est._flatten_pair(theta[:1], x[:5])
    assert (S, B) == (1, 5) and th.shape == (5, 4)


# ---------------------------------------------------------------- builders / configs
def test_builder_rejects_what_the_hip_path_does_not_implement():
    theta, x = linear_gaussian_data(50, 4, 7)
    with pytest.raises(ValueError, match="transform_to_unconstrained"):
        build_nsf(theta, x, z_score_x="transform_to_unconstrained")
    for bad in (-1, 5):         # more than four applications of the shared hidden layer: not built
        with pytest.raises(NotImplementedError):
            build_nsf(theta[:, :1], x, hidden_layers_spline_context=bad)
    assert build_nsf(theta[:, :1], x).net.hyper.ctx_mlp
    assert build_nsf(theta, x, hidden_layers_spline_context=3).net.hyper.hidden_layers_spline_context == 1   # theta-dim > 1: unused
    emb = build_nsf(theta, x, embedding_net=torch.nn.Linear(7, 3))       # embedded features feed the kernels
    assert emb.net.hyper.C == 3 and emb.condition_shape == torch.Size([7])
    assert not emb.net.z_score_x and isinstance(emb.embedding_net[0], torch.nn.Module)   # Standardize -> Linear
    with pytest.raises(ValueError, match="batch, features"):
        build_nsf(theta, x, embedding_net=torch.nn.Unflatten(1, (7, 1)))
    with pytest.raises(ValueError, match="Invalid z-scoring"):
        build_nsf(theta, x, z_score_y="bogus")


@pytest.mark.parametrize("zt", [None, "none", "independent", "structured"])
@pytest.mark.parametrize("zx", [None, "none", "independent", "structured"])
def test_z_score_flag_combinations_build(zt, zx):
    theta, x = linear_gaussian_data(64, 3, 5)
    est = build_nsf(theta, x, z_score_x=zt, z_score_y=zx)
    keys = est.net.nflows_state_dict().keys()
    assert any("_shift" in k for k in keys) == (zt in ("independent", "structured"))
    assert any("_embedding_net.0._mean" in k for k in keys) == (zx in ("independent", "structured"))
    if zt == "structured":
        assert est.net.zstats[3:6].unique().numel() == 1


def test_factory_and_config_behaviours():
    theta, x = linear_gaussian_data(64, 4, 7)
    with pytest.warns(UserWarning, match="Unknown kwargs"):
        build = posterior_nn("nsf", not_a_real_option=3)
    assert build(theta, x).net.hyper.hidden_features == 50
    with pytest.raises(NotImplementedError):
        posterior_nn("maf")(theta, x)
    cfg = NSFConfig(num_bins=8, hidden_features=32, z_score_input=None)
    est = cfg.build(theta, x)
    assert est.net.hyper.num_bins == 8 and not est.net.z_score_theta
    assert "num_bins=8" in repr(cfg) and "num_transforms" not in repr(cfg)
    with pytest.raises(ValueError):
        NSFConfig(z_score_input="bogus")
    with pytest.raises(Exception):
        cfg.num_bins = 3   # frozen


def test_flat_layout_and_nflows_state_dict_exchange_with_oracle():
    from oracle.nsf_oracle import NSFOracle

    theta, x = linear_gaussian_data(200, 5, 3)
    torch.manual_seed(1)
    oracle = NSFOracle(theta, x, hidden_features=32, num_transforms=3, num_bins=8)
    torch.manual_seed(1)
    est = build_nsf(theta, x, hidden_features=32, num_transforms=3, num_bins=8)
    sd_o, sd_e = oracle.state_dict(), est.net.nflows_state_dict()
    for k, v in sd_e.items():          # same seed => same nflows-order initialisation
        assert torch.equal(sd_o[k], v), k
    with torch.no_grad():
        for p in oracle.parameters():
            p.mul_(1.5)
    est.net.load_nflows_state_dict(oracle.state_dict())
    for k, v in est.net.nflows_state_dict().items():
        assert torch.equal(oracle.state_dict()[k], v)
    est2 = pickle.loads(pickle.dumps(est))
    assert torch.equal(est2.net.flat_params, est.net.flat_params)


# ---------------------------------------------------------------- trainer loop semantics
def _npe_with_oracle(n=600, D=2, C=2, **kw):
    theta, x = linear_gaussian_data(n, D, C)
    prior = torch.distributions.MultivariateNormal(torch.zeros(D), 0.1 * torch.eye(D))
    torch.manual_seed(2)
    inf = NPE(prior=prior, density_estimator=oracle_build_fn(hidden_features=16, num_transforms=2, num_bins=4, **kw),
              show_progress_bars=False)
    return inf.append_simulations(theta, x), theta, x


def test_estimator_arg_checks():
    with pytest.raises(TypeError):
        NPE(density_estimator=torch.nn.Linear(2, 2))
    with pytest.raises(TypeError):
        NPE(density_estimator=NSFConfig)
    with pytest.warns(FutureWarning):
        NPE(density_estimator="nsf")
    with pytest.raises(RuntimeError, match="append_simulations"):
        NPE(density_estimator=oracle_build_fn()).train()


def test_append_simulations_validation_and_invalid_rows():
    theta, x = linear_gaussian_data(100, 2, 2)
    inf = NPE(density_estimator=oracle_build_fn(), show_progress_bars=False)
    with pytest.raises(AssertionError, match="float32"):
        inf.append_simulations(theta.double(), x)
    with pytest.raises(AssertionError, match="must match"):
        inf.append_simulations(theta[:50], x)
    x = x.clone()
    x[3, 0] = float("nan")
    x[10, 1] = float("inf")
    inf.append_simulations(theta, x)
    assert inf.get_simulations()[0].shape[0] == 98
    # a proposal without a prior at initialization (npe_base.py:283-294)
    theta2, x2 = linear_gaussian_data(100, 2, 2)
    with pytest.raises(ValueError, match="did not pass a prior"):
        inf.append_simulations(theta2, x2, proposal=object())


def test_training_loop_split_sizes_and_early_stopping():
    inf, theta, x = _npe_with_oracle()
    est = inf.train(training_batch_size=100, max_num_epochs=3, stop_after_epochs=20)
    assert inf.train_indices.numel() == 540 and inf.val_indices.numel() == 60     # int(0.9*n) split
    s = inf.summary
    assert s["epochs_trained"][-1] == 4 and len(s["training_loss"]) == 4           # epoch <= max_num_epochs
    assert all(p.grad is None for p in est.parameters())                            # no grads left behind
    assert s["validation_loss"][-1] < s["validation_loss"][0]


def test_best_weights_restored_when_budget_exhausted():
    """base_test.py:261-293: after hitting max_num_epochs the kept weights score the best validation loss."""
    inf, theta, x = _npe_with_oracle(n=400)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.train(training_batch_size=90, learning_rate=5e-2, max_num_epochs=6)
    val = inf.val_indices
    with torch.no_grad():
        kept = est.loss(theta[val], x[val]).mean().item()
    assert abs(kept - min(inf.summary["validation_loss"])) < 5e-2 or kept <= min(inf.summary["validation_loss"]) + 5e-2
    assert inf._best_val_loss == min(inf.summary["validation_loss"])


def test_resume_training_keeps_split_and_epoch_count():
    inf, theta, x = _npe_with_oracle()
    inf.train(training_batch_size=100, max_num_epochs=1)
    tr = inf.train_indices.clone()
    e1 = inf.epoch
    inf.train(training_batch_size=100, max_num_epochs=3, resume_training=True)
    assert torch.equal(tr, inf.train_indices) and inf.epoch > e1


# ---------------------------------------------------------------- posterior / sampler
def test_direct_posterior_sampling_logprob_and_leakage_on_cpu_estimator():
    theta, x = linear_gaussian_data(500, 2, 2)
    torch.manual_seed(0)
    est = OracleEstimator(theta, x, hidden_features=16, num_transforms=2, num_bins=4)
    prior = BoxUniform(-0.3 * torch.ones(2), 0.3 * torch.ones(2))
    post = DirectPosterior(est, prior, device="cpu")
    with pytest.raises(ValueError, match="default has not been set"):
        post.sample((5,))
    post.set_default_x(torch.zeros(1, 2))
    s = post.sample((333,), show_progress_bars=False)
    assert s.shape == (333, 2) and bool(prior.support.check(s).all())
    lp = post.log_prob(torch.tensor([[0.0, 0.0], [5.0, 5.0]]))
    assert torch.isfinite(lp[0]) and lp[1] == float("-inf")
    acc = post.leakage_correction(torch.zeros(1, 2), num_rejection_samples=2000)
    assert 0.0 < acc.item() <= 1.0
    lp_un = post.log_prob(torch.zeros(1, 2), norm_posterior=False)
    assert torch.allclose(lp[0], lp_un[0] - torch.log(acc)[0], atol=1e-5)
    with pytest.raises(ValueError, match="batchsize == 1"):
        post.sample((3,), x=torch.zeros(2, 2))
    sb = post.sample_batched((7,), x=torch.zeros(3, 2), show_progress_bars=False)
    assert sb.shape == (7, 3, 2)
    lpb = post.log_prob_batched(sb, x=torch.zeros(3, 2), norm_posterior=False)
    assert lpb.shape == (7, 3)


def test_rejection_sampler_timeout_and_partial_return():
    from sbi_amd.samplers.rejection.rejection import accept_reject_sample

    def proposal(shape, condition):
        return torch.randn(shape[0], 1, 2)

    never = lambda t: torch.zeros(t.shape[:-1], dtype=torch.bool)   # noqa: E731
    with pytest.raises(RuntimeError, match="max_sampling_time"):
        accept_reject_sample(proposal, never, 10, max_sampling_time=0.05,
                             proposal_sampling_kwargs={"condition": torch.zeros(1, 2)})
    rare = lambda t: (t[..., 0] > 2.0)   # noqa: E731
    with pytest.warns(UserWarning, match="partial"):
        smp, _ = accept_reject_sample(proposal, rare, 10**7, max_sampling_time=0.2, return_partial_on_timeout=True,
                                      max_sampling_batch_size=1000,
                                      proposal_sampling_kwargs={"condition": torch.zeros(1, 2)})
    assert 0 < smp.shape[0] < 10**7 and bool((smp[..., 0] > 2.0).all())


def test_build_posterior_requires_direct_and_trained_net():
    inf, theta, x = _npe_with_oracle()
    with pytest.raises(ValueError):
        inf.build_posterior()
    inf.train(training_batch_size=100, max_num_epochs=1)
    with pytest.raises(NotImplementedError):
        inf.build_posterior(sample_with="vi")
    mcmc = inf.build_posterior(sample_with="mcmc", mcmc_parameters=dict(num_chains=4)).set_default_x(torch.zeros(2))
    assert mcmc.num_chains == 4 and mcmc.potential(torch.zeros(3, 2)).shape == (3,)
    pickle.loads(pickle.dumps(mcmc))                     # the parameter transform survives pickling
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # the sampler itself is a device kernel
        mcmc.sample((5,), init_strategy="proposal", show_progress_bars=False)
    post = inf.build_posterior().set_default_x(torch.zeros(2))
    assert post.sample((10,), show_progress_bars=False).shape == (10, 2)
    post2 = pickle.loads(pickle.dumps(post))            # save_and_load_test.py:23-43
    torch.manual_seed(0)
    a = post.sample((3,), show_progress_bars=False)
    torch.manual_seed(0)
    b = post2.sample((3,), show_progress_bars=False)
    assert torch.equal(a, b)


def test_npe_early_stopping_rule_matches_reference_trainer():
    """PosteriorEstimatorTrainer._converged replayed on the sequences tools/make_golden_npe_trainer.py fed to
    sbi's real NeuralInference._converged (base.py:1254-1284): decisions, counters, best loss, best weights held
    and weights restored on convergence are identical."""
    import os
    import types

    from sbi_amd.inference.trainers.npe.npe import PosteriorEstimatorTrainer

    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "npe_trainer_reference.pt"),
                      weights_only=False)
    assert len(gold) == 9
    for (name, stop), g in gold.items():
        net = torch.nn.Linear(1, 1, bias=False)
        with torch.no_grad():
            net.weight.fill_(-1.0)
        fake = types.SimpleNamespace(_neural_net=net, _val_loss=float("inf"), _best_val_loss=float("inf"),
                                     _epochs_since_last_improvement=0, _best_model_state_dict=None,
                                     _load_state=PosteriorEstimatorTrainer._load_state)
        for ep, v in enumerate(g["seq"]):
            c = PosteriorEstimatorTrainer._converged(fake, ep, stop)
            ref = g["trace"][ep]
            got = (bool(c), fake._epochs_since_last_improvement, fake._best_val_loss,
                   fake._best_model_state_dict["weight"].item(), net.weight.item())
            assert got == ref, (name, stop, ep, got, ref)
            if c:
                break
            with torch.no_grad():
                net.weight.fill_(float(ep + 1))
            fake._val_loss = v
        assert ep + 1 == len(g["trace"])


def test_tracker_receives_the_rounds_statistics():
    """`tracker=` (sbi/sbi_types.py:73-91) is honoured: the calls, tags and step numbers of
    trainers/base.py:1317-1385 (`_summarize`), once per train() call, for both rounds of a two-round run."""
    import warnings

    from sbi_amd.inference import NPE
    from tests.helpers import linear_gaussian_data
    from tests.oracle_adapter import oracle_build_fn

    class Recorder:
        log_dir = None

        def __init__(self):
            self.metrics, self.flushes = [], 0

        def log_metric(self, name, value, step=None):
            self.metrics.append((name, float(value), step))

        def log_metrics(self, metrics, step=None):
            for k, v in metrics.items():
                self.log_metric(k, v, step)

        def log_params(self, params):
            pass

        def add_figure(self, name, figure, step=None):
            pass

        def flush(self):
            self.flushes += 1

    theta, x = linear_gaussian_data(300, 2, 2)
    rec = Recorder()
    torch.manual_seed(0)
    inf = NPE(density_estimator=oracle_build_fn(hidden_features=8, num_transforms=1, num_bins=4), tracker=rec,
              show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.append_simulations(theta, x).train(training_batch_size=100, max_num_epochs=2)
        n1 = len(rec.metrics)
        inf.train(training_batch_size=100, max_num_epochs=4, resume_training=True)
    s = inf.summary
    first = rec.metrics[:n1]
    assert first[0] == ("epochs_trained", float(s["epochs_trained"][0]), 1)
    assert first[1] == ("best_validation_loss", float(s["best_validation_loss"][0]), 1)
    e0 = s["epochs_trained"][0]
    assert [m for m in first if m[0] == "validation_loss"] == \
        [("validation_loss", float(v), i) for i, v in enumerate(s["validation_loss"][:e0])]
    assert [m[2] for m in first if m[0] == "training_loss"] == list(range(e0))
    assert [m[2] for m in first if m[0] == "epoch_durations_sec"] == list(range(e0))
    # the second call logs only ITS epochs, offset by the epochs already trained (base.py:1358-1362)
    second = rec.metrics[n1:]
    assert [m[2] for m in second if m[0] == "validation_loss"] == list(range(e0, len(s["validation_loss"])))
    assert rec.flushes == 2


def test_context_spline_map_with_a_repeated_hidden_layer_exchanges_weights_with_the_reference_layout():
    """hidden_layers_spline_context = n (flow.py:346, 1456-1462): the reference puts the SAME Linear object at indices
    2, 4, ..., 2 n of `spline_predictor`, so its state_dict lists that one tensor n times and the output layer moves to
    index 2 + 2 n.  The flat buffer stores it once; the exchange emits / accepts every alias and refuses aliases that
    disagree (a checkpoint of a different architecture)."""
    from oracle.nsf_oracle import NSFOracle

    theta, x = linear_gaussian_data(60, 1, 3)
    torch.manual_seed(4)
    oracle = NSFOracle(theta, x, num_transforms=2, hidden_layers_spline_context=3)
    est = build_nsf(theta, x, num_transforms=2, hidden_layers_spline_context=3)
    assert est.net.hyper.param_count() == build_nsf(theta, x, num_transforms=2).net.hyper.param_count()   # shared weights
    sd_ref = oracle.state_dict()
    pre = "net._transform._transforms.1.transform_net.spline_predictor."
    assert {k for k in sd_ref if k.startswith(pre)} == {pre + f"{i}.{w}" for i in (0, 2, 4, 6, 8) for w in ("weight", "bias")}
    est.net.load_nflows_state_dict(sd_ref)
    sd = est.net.nflows_state_dict()
    missing, unexpected = oracle.load_state_dict(sd, strict=False)    # same parameter keys, aliases included
    assert not unexpected and all(k.endswith("_features") for k in missing)   # (only the coupling index buffers)
    for k, v in sd_ref.items():
        if not k.endswith("_features"):
            assert torch.equal(sd[k], v), k
    broken = dict(sd_ref)
    broken[pre + "4.weight"] = broken[pre + "4.weight"] + 1.0
    with pytest.raises(ValueError, match="repeated hidden layer"):
        est.net.load_nflows_state_dict(broken)


def test_mcmc_posterior_pickles_after_it_has_sampled():
    """`sample()` leaves the chains' potential closure on the posterior (`potential_`, and inside `_posterior_sampler`):
    neither is part of the pickled state, and pickling does not touch the live object (tests/save_and_load_test.py:23-45;
    the device test is tests/test_npe_gpu.py::test_device_posterior_survives_pickling_...[mcmc])."""
    import pickle

    from sbi_amd.inference.posteriors.mcmc_posterior import MCMCPosterior

    post = MCMCPosterior.__new__(MCMCPosterior)
    closure = lambda u: u          # noqa: E731  (what sample() stores: a local function)
    post.__dict__.update(dict(potential_=closure, _posterior_sampler=type("S", (), {})(), method="slice_np_vectorized"))
    post._posterior_sampler.log_prob_fn = closure
    before = {k: id(v) for k, v in vars(post).items()}
    clone = pickle.loads(pickle.dumps(post))
    assert {k: id(v) for k, v in vars(post).items()} == before
    assert clone.potential_ is None and clone._posterior_sampler is None and clone.method == "slice_np_vectorized"
