"""Checkpoints are interchangeable with the reference's through the PLAIN calls: `state_dict()` of the estimator speaks
nflows' key names (what sbi's trainer snapshots with `deepcopy(neural_net.state_dict())`,
trainers/base.py:1275-1281, and what users persist, tests/save_and_load_test.py:23-43), `load_state_dict()` reads them.
The oracle carries nflows' module tree, so `oracle.state_dict()` stands in for a checkpoint written by real sbi."""
import copy
import io

import pytest
import torch

from tests.helpers import make_inputs, matched_pair

CASES = [dict(D=3, C=2), dict(D=1, C=3, hidden_layers_spline_context=3),
         dict(D=4, C=2, z_score_theta="none", z_score_x="none"), dict(D=10, C=10)]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_keys_and_values_match_the_reference_layout(kw):
    oracle, est, _, _ = matched_pair(device=None, **kw)
    sd_o, sd_e = oracle.state_dict(), est.state_dict()
    assert set(sd_o) == set(sd_e)
    for k, v in sd_o.items():
        assert sd_e[k].dtype == v.dtype and torch.equal(sd_e[k], v), k
    # both directions with the standard, strict call
    oracle2, est2, _, _ = matched_pair(device=None, seed=7, **kw)
    assert not torch.equal(est2.net.flat_params, est.net.flat_params)
    est2.load_state_dict(sd_o, strict=True)
    assert torch.equal(est2.net.flat_params, est.net.flat_params) and torch.equal(est2.net.zstats, est.net.zstats)
    oracle2.load_state_dict(sd_e, strict=True)
    for (ka, a), (kb, b) in zip(oracle.state_dict().items(), oracle2.state_dict().items()):
        assert ka == kb and torch.equal(a, b)


def test_torch_save_load_round_trip_and_native_form():
    _, est, _, _ = matched_pair(device=None, D=3, C=2)
    buf = io.BytesIO()
    torch.save(est.state_dict(), buf)
    buf.seek(0)
    _, other, _, _ = matched_pair(device=None, D=3, C=2, seed=9)
    other.load_state_dict(torch.load(buf))
    assert torch.equal(other.net.flat_params, est.net.flat_params)
    # the two-tensor form of earlier versions still loads; deepcopy of a state_dict still restores
    native = est.net.native_state_dict()
    assert set(native) == {"flat_params", "zstats"}
    other.net.load_state_dict(native)
    snap = copy.deepcopy(est.state_dict())
    with torch.no_grad():
        est.net.flat_params.add_(1.0)
    est.load_state_dict(snap)
    assert torch.equal(other.net.flat_params, est.net.flat_params)


def test_incomplete_checkpoint_is_refused():
    oracle, est, _, _ = matched_pair(device=None, D=3, C=2)
    bad = dict(oracle.state_dict())
    bad.pop(next(k for k in bad if "final_layer.weight" in k))
    with pytest.raises(RuntimeError):
        est.load_state_dict(bad)
    wrong = dict(oracle.state_dict())
    k = next(k for k in wrong if "initial_layer.weight" in k)
    wrong[k] = wrong[k][:, :-1]
    with pytest.raises(RuntimeError):
        est.load_state_dict(wrong)


@pytest.mark.gpu
def test_reference_checkpoint_loads_on_the_device_and_evaluates():
    oracle, _, _, _ = matched_pair(device=None, D=10, C=10)
    _, est, _, _ = matched_pair(D=10, C=10, seed=11)          # different weights, on the ROCm device
    est.load_state_dict(oracle.state_dict())                  # CPU tensors into device parameters: the plain call
    theta, x = make_inputs(4096, 10, 10)
    with torch.no_grad():
        ref = oracle.log_prob(theta, x)[0]
    got = est.log_prob(theta.cuda(), x.cuda())[0].cpu()
    assert (got - ref).abs().max() <= 1e-5 * (1 + ref.abs().max())
    # and back: the device estimator's checkpoint into an oracle
    oracle2, _, _, _ = matched_pair(device=None, D=10, C=10, seed=13)
    oracle2.load_state_dict({k: v.cpu() for k, v in est.state_dict().items()}, strict=True)
    with torch.no_grad():
        assert torch.equal(oracle2.log_prob(theta, x)[0], ref)
