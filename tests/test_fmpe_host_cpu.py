"""Host-side pieces of the FMPE path that need no GPU: the adaptive ODE integrator, parameter layout and
state-dict exchange with the reference naming, refusal of unsupported options and of CPU tensors."""

import math

import pytest
import torch

from sbi_amd.samplers.ode_solvers import odeint_dopri5


def test_dopri5_linear_and_nonautonomous_both_directions():
    y0 = torch.tensor([[1.0, -2.0], [0.5, 3.0]], dtype=torch.float64)
    y = odeint_dopri5(lambda t, y: -y, y0, 0.0, 1.0, atol=1e-9, rtol=1e-8)
    assert (y - y0 * math.exp(-1.0)).abs().max() < 1e-7
    back = odeint_dopri5(lambda t, y: -y, y, 1.0, 0.0, atol=1e-9, rtol=1e-8)
    assert (back - y0).abs().max() < 1e-6
    # dy/dt = t * y  ->  y(1) = y0 * exp(1/2)
    y = odeint_dopri5(lambda t, y: t * y, y0, 0.0, 1.0, atol=1e-9, rtol=1e-8)
    assert (y - y0 * math.exp(0.5)).abs().max() < 1e-6
    # sbi's tolerances give ~1e-5 accuracy
    y = odeint_dopri5(lambda t, y: torch.sin(5 * t) * y, y0.float(), 1.0, 0.0)
    ref = y0 * math.exp((math.cos(5.0) - 1.0) / 5.0)
    assert (y.double() - ref).abs().max() < 1e-4
    assert torch.equal(odeint_dopri5(lambda t, y: y, y0, 0.3, 0.3), y0)


def test_parameter_layout_matches_c_abi_and_reference_names():
    from sbi_amd import _lib
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import FMPEHyper, build_flow_matching_estimator
    from oracle.fmpe_oracle import FMPEOracle

    lib = _lib.load()
    for kw in [dict(D=5, C=3), dict(D=50, C=50), dict(D=3, C=4, hidden_features=48, num_layers=2)]:
        h = FMPEHyper(**kw)
        assert lib.sbi_amd_fmpe_param_count(h.c_config()) == h.param_count()
        assert lib.sbi_amd_fmpe_packed_floats(h.c_config()) > h.param_count()
    assert FMPEHyper(D=5, C=3).param_count() == 76405      # sbi's default net on the golden case
    assert lib.sbi_amd_fmpe_param_count(FMPEHyper(D=5, C=3, hidden_features=200).c_config()) == _lib.E_UNSUPPORTED
    assert lib.sbi_amd_fmpe_param_count(FMPEHyper(D=5, C=3, time_embedding_dim=33).c_config()) == _lib.E_UNSUPPORTED
    torch.manual_seed(0)
    theta, x = torch.randn(64, 4) * 2 + 1, torch.randn(64, 6)
    est = build_flow_matching_estimator(theta, x, hidden_features=32, num_layers=2)
    sd = est.net.reference_state_dict()
    o = FMPEOracle(4, 6, H=32, L=2)
    o.load_reference_state_dict(sd)            # every key the oracle (= reference naming) needs is present
    assert torch.allclose(sd["mean_0"], theta.mean(0)) and torch.allclose(sd["std_0"], theta.std(0))
    assert torch.count_nonzero(sd["net.output_layer.weight"]) == 0      # zero-initialised like sbi
    est2 = build_flow_matching_estimator(theta, x, hidden_features=32, num_layers=2)
    est2.net.load_reference_state_dict(sd)
    assert torch.equal(est2.net.flat_params, est.net.flat_params)


def test_unsupported_options_and_cpu_tensors_are_refused():
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    theta, x = torch.randn(32, 3), torch.randn(32, 2)
    with pytest.raises(NotImplementedError):
        build_flow_matching_estimator(theta, x, net="transformer")
    with pytest.raises(NotImplementedError):
        build_flow_matching_estimator(theta, x, gaussian_baseline=True)
    est = build_flow_matching_estimator(theta, x)
    with pytest.raises(RuntimeError, match="no CPU fallback|ROCm"):
        est.loss(theta, x)
    with pytest.raises(RuntimeError, match="no CPU fallback|ROCm"):
        est(theta, x, torch.rand(32))
    from sbi_amd.inference import FMPE

    with pytest.raises(RuntimeError, match="ROCm"):
        FMPE(prior=None, device="cpu").append_simulations(theta, x).train(max_num_epochs=1)


def test_trainer_loop_bookkeeping_with_oracle_backed_estimator():
    """Epoch loop of the FMPE trainer on a CPU stand-in estimator: EMA-smoothed summaries, validation at fixed
    times (tensor or count), resume_training, calibration kernel, early-stopping rule."""
    import warnings

    from sbi_amd.inference import FMPE
    from tests.helpers import linear_gaussian_data
    from tests.oracle_adapter import oracle_vf_build_fn

    theta, x = linear_gaussian_data(300, 3, 2)
    torch.manual_seed(0)
    inf = FMPE(vf_estimator=oracle_vf_build_fn(H=16, L=1, E=8), show_progress_bars=False)
    inf.append_simulations(theta, x)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.train(training_batch_size=64, max_num_epochs=3, validation_times=torch.tensor([0.2, 0.5, 0.8]),
                  ema_loss_decay=0.25)
        s = inf.summary
        assert len(s["training_loss"]) == len(s["validation_loss"]) == 4 and s["epochs_trained"] == [4]
        assert all(torch.isfinite(torch.tensor(s["validation_loss"])))
        # resume: epochs continue, split is kept
        split = inf.train_indices.clone()
        inf.train(training_batch_size=64, max_num_epochs=5, resume_training=True, validation_times=3,
                  calibration_kernel=lambda xx: torch.ones(xx.shape[0]) * 2.0)
        assert torch.equal(split, inf.train_indices) and inf.epoch == 6
    # the convergence rule: an epoch is only fruitless when more than two running stds above the best
    inf._summary["validation_loss"] = [1.0, 0.9, 1.1, 1.0] * 3
    inf._best_val_loss, inf._val_loss, inf._epochs_since_last_improvement = 0.5, 0.55, 0
    assert inf._converged(epoch=7, stop_after_epochs=4) is False and inf._epochs_since_last_improvement == 0
    inf._val_loss = 2.0
    inf._converged(epoch=8, stop_after_epochs=4)
    assert inf._epochs_since_last_improvement == 1
    with pytest.raises(NotImplementedError):
        inf.append_simulations(theta, x, proposal=object())


def test_estimator_and_posterior_pickle_round_trip():
    """save_and_load_test.py of the reference pickles estimators / posteriors; joblib workers do the same."""
    import pickle
    from copy import deepcopy

    from sbi_amd.inference.posteriors.vector_field_posterior import VectorFieldPosterior
    from sbi_amd.neural_nets.estimators.flowmatching_estimator import build_flow_matching_estimator

    theta, x = torch.randn(40, 3), torch.randn(40, 2)
    est = build_flow_matching_estimator(theta, x, hidden_features=32, num_layers=2)
    est2 = pickle.loads(pickle.dumps(est))
    assert torch.equal(est2.net.flat_params, est.net.flat_params) and torch.equal(est2.net.zstats, est.net.zstats)
    assert est2.net.hyper == est.net.hyper and est2.input_shape == est.input_shape
    est3 = deepcopy(est)
    assert torch.equal(est3.net.flat_params, est.net.flat_params)
    prior = torch.distributions.Independent(torch.distributions.Normal(torch.zeros(3), torch.ones(3)), 1)
    post = VectorFieldPosterior(est, prior, device="cpu")
    post2 = pickle.loads(pickle.dumps(post))
    assert torch.equal(post2.vector_field_estimator.net.flat_params, est.net.flat_params)
    sd = est.state_dict()
    est4 = build_flow_matching_estimator(theta + 1.0, x, hidden_features=32, num_layers=2)
    est4.load_state_dict(sd)
    assert torch.equal(est4.net.zstats, est.net.zstats)
