"""The device-resident Dormand-Prince stepper (csrc/ode.hip behind sbi_amd_dopri5_*) against the same method with a
host-side controller (torch ops) and against a tight fp64 solve, on right-hand sides with known behaviour."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rhs(t, y):                      # t: 1-element tensor (what the velocity kernel is handed)
    return torch.sin(5.0 * t) * y - 0.7 * torch.tanh(2.0 * y)     # Lipschitz: well posed in both time directions


@pytest.mark.parametrize("span", [(1.0, 0.0), (0.0, 2.0), (0.25, 0.25)])
@pytest.mark.parametrize("shape", [(1000, 7), (3, 1), (257, 50)])
def test_device_stepper_matches_host_controller_and_tight_solve(span, shape):
    from sbi_amd.samplers.ode_solvers.dopri5 import _odeint_device, _odeint_host, odeint_dopri5

    t0, t1 = span
    g = torch.Generator().manual_seed(3)
    y0 = torch.randn(*shape, generator=g).cuda()
    got = odeint_dopri5(_rhs, y0, t0, t1)
    assert got.shape == y0.shape and got.dtype == torch.float32 and torch.isfinite(got).all()
    if t0 == t1:
        assert torch.equal(got, y0)
        return
    host = _odeint_host(_rhs, y0, t0, t1, 1e-6, 1e-5, 10_000, 0.05)
    tight = _odeint_host(_rhs, y0.double(), t0, t1, 1e-11, 1e-10, 100_000, 0.01)
    e_dev = (got.double() - tight).abs().max().item()
    e_host = (host.double() - tight).abs().max().item()
    print(f"span {span} shape {shape}: device vs tight {e_dev:.2e}, host-controller vs tight {e_host:.2e}, "
          f"device vs host {(got - host).abs().max().item():.2e}")
    assert e_dev <= 5e-5 * max(1.0, tight.abs().max().item())
    assert e_dev <= 3.0 * e_host + 1e-5
    # same controller: the two take the same sequence of attempts up to fp32 rounding of the error norm
    assert (got - host).abs().max().item() <= 2e-5 * max(1.0, tight.abs().max().item())
    # direct call leaves the input untouched
    again = _odeint_device(_rhs, y0, t0, t1, 1e-6, 1e-5, 10_000, 0.05)
    assert torch.equal(again, got)
    # the same attempts replayed from a captured HIP graph
    replayed = _odeint_device(_rhs, y0, t0, t1, 1e-6, 1e-5, 10_000, 0.05, use_graph=True)
    assert torch.equal(replayed, got)


def test_device_stepper_rejects_and_recovers_on_a_stiff_start():
    """first_step far too large for the dynamics: attempts must be rejected (h shrinks) and the solve still converge."""
    from sbi_amd.samplers.ode_solvers.dopri5 import _odeint_host, odeint_dopri5

    y0 = torch.full((64, 4), 2.0).cuda()
    f = lambda t, y: -40.0 * y
    got = odeint_dopri5(f, y0, 0.0, 1.0, first_step=1.0)
    tight = _odeint_host(f, y0.double(), 0.0, 1.0, 1e-12, 1e-11, 100_000, 0.001)
    assert (got.double() - tight).abs().max().item() <= 1e-5
