"""The FMPE oracle (oracle/fmpe_oracle.py) against outputs of the real sbi classes
(tests/golden/fmpe_reference.pt, written by tools/make_golden_fmpe.py): per-row CFM loss, parameter
gradients of the mean loss and the velocity handed to ODE solvers.  Tolerance: fp32 round-off (1e-5 rel)."""

import os

import pytest
import torch

from oracle.fmpe_oracle import FMPEOracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fmpe_reference.pt")


def load_case(name):
    g = torch.load(GOLD, weights_only=False)[name]
    kw = g["kw"]
    o = FMPEOracle(g["D"], g["C"], H=kw.get("hidden_features", 100), L=kw.get("num_layers", 5))
    o.load_reference_state_dict(g["state"])
    return g, o


@pytest.mark.parametrize("name", ["default_D5_C3", "H48_L2_D3_C4"])
def test_loss_and_gradients_match_reference(name):
    g, o = load_case(name)
    losses = o.loss(g["theta"], g["x"], g["times"], g["noise"])
    assert (losses - g["losses"]).abs().max() <= 1e-5 * g["losses"].abs().max()
    o.zero_grad()
    losses.mean().backward()
    for k, ref in g["grads"].items():
        got = o.p[k.replace(".", "/")].grad
        assert (got - ref).abs().max() <= 2e-5 * ref.abs().max() + 1e-7, k


@pytest.mark.parametrize("name", ["default_D5_C3", "H48_L2_D3_C4"])
def test_velocity_matches_reference(name):
    g, o = load_case(name)
    with torch.no_grad():
        v = o.velocity(g["theta_q"], g["x"][:1], g["tq"])
    assert (v - g["vel"]).abs().max() <= 1e-5 * g["vel"].abs().max()
