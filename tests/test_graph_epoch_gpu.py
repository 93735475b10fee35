"""`NPE.train()` with its epochs captured as HIP graphs (opt-in: SBI_AMD_GRAPH_EPOCH=1) (the reference's loop is
sbi/inference/trainers/base.py:1150-1225) against the same call on the eager pipelined loop: same seeds => the same
sampler orders (the kernel derives the epoch key from the device clock exactly as the host does), the same steps, the
same early-stopping decisions.  The only arithmetic that differs is Adam's bias correction (1 - beta^step evaluated
by the device's `pow` instead of the host's): identical after rounding to fp32 except for a last-bit case now and then,
so losses are compared to 1e-6 relative and the exact-equality share is reported."""
import warnings

import pytest
import torch
from torch.distributions import MultivariateNormal

from sbi_amd.inference import NPE
from sbi_amd.neural_nets import NSFConfig
from sbi_amd.simulators.linear_gaussian import diagonal_linear_gaussian

pytestmark = pytest.mark.gpu


def _run(monkeypatch, graph, dim, n, batch, **kw):
    monkeypatch.setenv("SBI_AMD_GRAPH_EPOCH", "1" if graph else "0")
    torch.manual_seed(0)
    prior = MultivariateNormal(torch.zeros(dim, device="cuda"), 0.1 * torch.eye(dim, device="cuda"))
    theta = prior.sample((n,)).cpu()
    x = diagonal_linear_gaussian(theta, std=0.1**0.5)
    torch.manual_seed(1)
    inf = NPE(prior=prior, density_estimator=NSFConfig(), device="cuda", show_progress_bars=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est = inf.append_simulations(theta, x).train(training_batch_size=batch, **kw)
    return inf, est


@pytest.mark.parametrize("dim,n,batch,kw", [
    (2, 2000, 200, dict(max_num_epochs=25, stop_after_epochs=25)),          # cooperative kernels, 9 steps per epoch
    (10, 20000, 4096, dict(max_num_epochs=20, stop_after_epochs=20)),
    (10, 100000, 65536, dict(max_num_epochs=30, stop_after_epochs=30)),     # BASELINE configs[1]: throughput kernels
    (10, 20000, 1000, dict(stop_after_epochs=3)),                           # early stopping: restore + discarded epoch
])
def test_graph_epochs_reproduce_the_eager_loop(monkeypatch, dim, n, batch, kw):
    inf_e, est_e = _run(monkeypatch, False, dim, n, batch, **kw)
    inf_g, est_g = _run(monkeypatch, True, dim, n, batch, **kw)
    assert getattr(inf_e, "_graph_epochs", 0) == 0
    assert getattr(inf_g, "_graph_epochs", 0) >= inf_g.summary["epochs_trained"][-1] - 2, "the graph path did not run"
    assert inf_g.summary["epochs_trained"] == inf_e.summary["epochs_trained"]
    for key in ("training_loss", "validation_loss"):
        a, b = torch.tensor(inf_e.summary[key]), torch.tensor(inf_g.summary[key])
        rel = ((a - b).abs() / (1 + a.abs())).max().item()
        print(f"{key}: {len(a)} epochs, max rel diff {rel:.2e}, exactly equal {(a == b).float().mean().item():.1%}")
        assert rel <= 1e-5
    assert abs(inf_g.summary["best_validation_loss"][-1] - inf_e.summary["best_validation_loss"][-1]) <= 1e-5
    d = (est_g.net.flat_params - est_e.net.flat_params).abs().max().item()
    print(f"final weights: max abs diff {d:.2e}")
    assert d <= 5e-4
    # the trained estimator answers as any other (the packed-image cache was reset after the replays)
    th = torch.zeros(7, dim, device="cuda")
    xx = torch.zeros(7, dim, device="cuda")
    assert torch.isfinite(est_g.log_prob(th.unsqueeze(0), xx)).all()
    s = inf_g.build_posterior().set_default_x(torch.zeros(1, dim)).sample((100,), show_progress_bars=False)
    assert s.shape == (100, dim) and torch.isfinite(s).all()


def test_resume_training_after_graph_epochs(monkeypatch):
    inf, _ = _run(monkeypatch, True, 2, 2000, 200, max_num_epochs=5, stop_after_epochs=50)
    steps = inf._stepper.step_count
    assert steps == 6 * 9 and int(inf._stepper.clock[1]) == steps     # epochs 0 ... 5 (max_num_epochs is inclusive)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        inf.train(training_batch_size=200, max_num_epochs=9, stop_after_epochs=50, resume_training=True)
    assert inf.summary["epochs_trained"][-1] == 10 and inf._stepper.step_count == 10 * 9
    assert len(inf.summary["training_loss"]) == 10
