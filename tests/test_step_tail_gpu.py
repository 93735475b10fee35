"""The table-driven re-pack at the end of a training step (csrc/step_tail.hip).

Contract (include/sbi_amd_nsf.h): after `sbi_amd_nsf_table_pack` the image(s) named in the table are BIT-IDENTICAL to
what `sbi_amd_nsf_pack_images(images)` writes from the same parameters -- the pack kernels every other test of the
training pass was written against.  Replaces what nflows re-derives inside every forward call
(LULinear._create_lower_upper and the .t() views, nflows transforms/lu.py) after the optimizer step of sbi's loop body
(sbi/inference/trainers/base.py:1181-1187)."""
import os

import pytest
import torch

from sbi_amd import _lib
from sbi_amd.inference.trainers.fused import FusedTrainStep
from sbi_amd.neural_nets.estimators.nsf_flow import NSFHyper, packed_weights
from sbi_amd.neural_nets.net_builders.flow import build_nsf

gpu = pytest.mark.gpu

SHAPES = {
    "default": dict(D=10, C=10),
    "D2-C2": dict(D=2, C=2),
    "theta-dim-1": dict(D=1, C=3),
    "16-bins-hidden-64": dict(D=5, C=7, hidden_features=64, num_bins=16, num_transforms=3),
    "one-block-4-bins": dict(D=7, C=4, num_blocks=1, num_bins=4),
    "wide-hidden-100": dict(D=10, C=10, hidden_features=100),
    "theta-dim-20-generic-pass": dict(D=20, C=6),
}


def _two_call(lib, cfg, p, g, m, v, step, scratch, packed, mask, max_norm):
    dev = p.device
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_adam_clip_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), step, 5e-4, 0.9,
                                        0.999, 1e-8, max_norm, _lib.ptr(scratch), _lib.current_stream(dev))
        assert rc == 0
        rc = lib.sbi_amd_nsf_pack_images(cfg, _lib.ptr(p), _lib.ptr(packed), mask, _lib.current_stream(dev))
        assert rc == 0


@gpu
@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("mask", [1, 2, 3])
def test_table_pack_is_bit_identical_to_the_pack_kernels(name, mask):
    lib = _lib.load()
    h = NSFHyper(**SHAPES[name])
    cfg = h.c_config()
    P = h.param_count()
    n_img = lib.sbi_amd_nsf_packed_floats(cfg)
    has_coop = lib.sbi_amd_nsf_coop_selfcheck(cfg) == 0
    wide = h.hidden_features > 64
    if (mask & 2) and not has_coop:
        pytest.skip("no cooperative image for this shape")
    if (mask & 1) and wide:
        pytest.skip("hidden > 64 has no throughput image")
    dev = torch.device("cuda:0")
    g0 = torch.Generator(device="cpu").manual_seed(3)
    p_ref = (0.3 * torch.randn(P, generator=g0)).to(dev)
    p_fus = p_ref.clone()
    m_ref, v_ref = torch.zeros_like(p_ref), torch.zeros_like(p_ref)
    m_fus, v_fus = torch.zeros_like(p_ref), torch.zeros_like(p_ref)
    s_ref, s_fus = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    img_ref = torch.zeros(int(n_img), device=dev)
    img_fus = torch.zeros(int(n_img), device=dev)
    mp = torch.zeros(int(lib.sbi_amd_nsf_step_map_ints(cfg)), dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.sbi_amd_nsf_step_map_workspace_floats(cfg)), device=dev)
    with torch.cuda.device(dev):
        rc = lib.sbi_amd_nsf_build_step_map(cfg, mask, _lib.ptr(p_fus), _lib.ptr(img_fus), _lib.ptr(mp), _lib.ptr(ws),
                                            _lib.current_stream(dev))
    assert rc == 0, f"build_step_map -> {rc}"
    hdr = mp[:8].cpu().tolist()
    assert hdr[1] == mask and hdr[2] == P
    # the build leaves `packed` fully packed from `params`
    with torch.cuda.device(dev):
        assert lib.sbi_amd_nsf_pack_images(cfg, _lib.ptr(p_ref), _lib.ptr(img_ref), mask, _lib.current_stream(dev)) == 0
    assert torch.equal(img_ref.view(torch.int32), img_fus.view(torch.int32))
    for step in range(1, 6):
        # big gradients first (the clip is active), small ones later (coefficient exactly 1)
        g = (torch.randn(P, generator=g0) * (0.5 if step < 3 else 1e-3)).to(dev)
        max_norm = 5.0 if step != 4 else 0.0                   # 0: clipping off
        _two_call(lib, cfg, p_ref, g, m_ref, v_ref, step, s_ref, img_ref, mask, max_norm)
        with torch.cuda.device(dev):
            rc = lib.sbi_amd_adam_clip_step(_lib.ptr(p_fus), _lib.ptr(g), _lib.ptr(m_fus), _lib.ptr(v_fus), P, step, 5e-4,
                                            0.9, 0.999, 1e-8, max_norm, _lib.ptr(s_fus), _lib.current_stream(dev))
            assert rc == 0
            rc = lib.sbi_amd_nsf_table_pack(cfg, _lib.ptr(p_fus), _lib.ptr(img_fus), _lib.ptr(mp),
                                            _lib.current_stream(dev))
        assert rc == 0
        for a, b, what in ((p_ref, p_fus, "params"), (m_ref, m_fus, "exp_avg"), (v_ref, v_fus, "exp_avg_sq"),
                           (img_ref, img_fus, "image"), (s_ref[:1], s_fus[:1], "norm")):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (name, mask, step, what,
                                                                           int((a != b).sum()))
    assert torch.isfinite(img_fus).all()


@gpu
@pytest.mark.parametrize("batch", [200, 8192, 20000])
def test_training_steps_with_and_without_the_table_pack_agree_bit_for_bit(batch, monkeypatch):
    """FusedTrainStep end to end (both kernel families), and the packed-image bookkeeping around it: after steps at
    one batch size, a log_prob call that reads the OTHER image (and the sampling direction's explicit inverses) must
    see the new parameters."""
    torch.manual_seed(0)
    theta = torch.randn(30000, 10)
    x = theta + 0.3 * torch.randn(30000, 10)
    runs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SBI_AMD_FUSED_TAIL", flag)
        torch.manual_seed(1)
        est = build_nsf(theta, x).cuda()
        st = FusedTrainStep(est)
        tb, xb = theta[:batch].cuda(), x[:batch].cuda()
        losses = [st.step(tb, xb).clone() for _ in range(4)]
        used = st.__dict__.get("_step_maps")
        assert (used is not None and len(used) == 1) == (flag == "1")      # one table: one image is trained on
        other = 20000 if batch <= 8192 else 200            # the other kernel family's image
        lp_other = est.log_prob(theta[:other].cuda(), x[:other].cuda())
        smp = est.sample_from_noise(torch.randn(64, 10, generator=torch.Generator().manual_seed(5)).cuda(), xb[:64])
        losses.append(st.step(tb, xb).clone())             # and back: the training image again
        runs[flag] = (est.net.flat_params.detach().clone(), st.exp_avg.clone(), losses, lp_other, smp,
                      packed_weights(est.net).clone())
    a, b = runs["1"], runs["0"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for la, lb in zip(a[2], b[2]):
        assert torch.equal(la, lb)
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert torch.equal(a[5].view(torch.int32), b[5].view(torch.int32))     # every image, inverses included


@gpu
def test_step_map_refuses_what_it_cannot_verify():
    lib = _lib.load()
    h = NSFHyper(D=10, C=10)
    cfg = h.c_config()
    dev = torch.device("cuda:0")
    p = torch.zeros(h.param_count(), device=dev)
    img = torch.zeros(int(lib.sbi_amd_nsf_packed_floats(cfg)), device=dev)
    mp = torch.zeros(int(lib.sbi_amd_nsf_step_map_ints(cfg)), dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.sbi_amd_nsf_step_map_workspace_floats(cfg)), device=dev)
    with torch.cuda.device(dev):
        st = _lib.current_stream(dev)
        assert lib.sbi_amd_nsf_build_step_map(cfg, 0, _lib.ptr(p), _lib.ptr(img), _lib.ptr(mp), _lib.ptr(ws), st) == _lib.E_BADARG
        assert lib.sbi_amd_nsf_build_step_map(cfg, 4, _lib.ptr(p), _lib.ptr(img), _lib.ptr(mp), _lib.ptr(ws), st) == _lib.E_BADARG
        big_eps = NSFHyper(D=10, C=10).c_config()
        big_eps.lu_eps = 0.5        # the probe could not tell softplus(u) + eps from a copy any more: refused, not guessed
        assert lib.sbi_amd_nsf_build_step_map(big_eps, 1, _lib.ptr(p), _lib.ptr(img), _lib.ptr(mp), _lib.ptr(ws), st) == _lib.E_UNSUPPORTED
        assert lib.sbi_amd_nsf_table_pack(cfg, _lib.ptr(p), _lib.ptr(img), None, st) == _lib.E_BADARG


# ---------------------------------------------------------------------------------------------- the norm rider
@gpu
@pytest.mark.parametrize("name,rows", [("default", 200), ("default", 8192), ("default", 20000), ("theta-dim-1", 300),
                                       ("16-bins-hidden-64", 5000), ("wide-hidden-100", 700),
                                       ("theta-dim-20-generic-pass", 400)])
def test_gradient_reduction_leaves_the_squared_norm_of_the_gradient(name, rows):
    """`sbi_amd_nsf_train_sqnorm_parts`: the partial sums the reduction kernels leave in the workspace add up to
    |grad_out|^2 (what clip_grad_norm_ needs, trainers/base.py:1182-1186), and `sbi_amd_adam_clip_step_parts` takes the
    step `sbi_amd_adam_clip_step` takes (same clip coefficient up to the association of the sum).  The generic training
    pass leaves none and says so."""
    import ctypes

    from sbi_amd.neural_nets.estimators.nsf_flow import NSFNet, loss_fwd_bwd, train_workspace
    from tests.helpers import linear_gaussian_data

    lib = _lib.load()
    kw = dict(SHAPES[name])
    D, C = kw.pop("D"), kw.pop("C")
    theta, x = linear_gaussian_data(max(rows, 64), D, C)
    torch.manual_seed(2)
    est = build_nsf(theta, x, **kw).cuda()
    net = est.net
    assert type(net) is NSFNet
    th, xx = theta[:rows].cuda(), x[:rows].cuda()
    grad = torch.zeros_like(net.flat_params.data)
    ws = train_workspace(net, rows, th.device)
    loss_fwd_bwd(net, th, xx, None, 37.0 / rows, grad, workspace=ws)       # a weight that makes the clip bite
    n_parts = ctypes.c_int64(-1)
    ptr = lib.sbi_amd_nsf_train_sqnorm_parts(net.hyper.c_config(), rows, _lib.ptr(ws), ctypes.byref(n_parts))
    if name == "theta-dim-20-generic-pass":
        assert not ptr and n_parts.value == 0
        return
    assert ptr and n_parts.value >= 1
    off = (ptr - ws.data_ptr()) // 4
    parts = ws[off:off + n_parts.value]
    total = float(parts.double().sum())
    want = float((grad.double() ** 2).sum())
    assert abs(total - want) <= 2e-6 * want, (total, want)
    assert want ** 0.5 > 5.0                                               # the clip IS active in this comparison
    P = grad.numel()
    outs = []
    for use_parts in (True, False):
        p = net.flat_params.data.clone()
        m, v, s = torch.zeros_like(p), torch.zeros_like(p), torch.zeros(256, device=p.device)
        with torch.cuda.device(p.device):
            st = _lib.current_stream(p.device)
            for step in (1, 2):
                if use_parts:
                    rc = lib.sbi_amd_adam_clip_step_parts(_lib.ptr(p), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), P, step, 5e-4,
                                                          0.9, 0.999, 1e-8, 5.0, ptr, n_parts.value, _lib.ptr(s), st)
                else:
                    rc = lib.sbi_amd_adam_clip_step(_lib.ptr(p), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), P, step, 5e-4, 0.9,
                                                    0.999, 1e-8, 5.0, _lib.ptr(s), st)
                assert rc == 0
        outs.append((p, m, v, float(s[0])))
    assert abs(outs[0][3] - outs[1][3]) <= 1e-6 * outs[1][3] and abs(outs[1][3] - want ** 0.5) <= 1e-5 * want ** 0.5
    # the two norms differ in the association of a float32 sum (up to ~1e-6 relative with a thousand partial sums), the
    # clip coefficient with them, the second moment (quadratic in the scaled gradient) twice as much
    # (a parameter that an update of lr = 5e-4 carries across zero has no relative accuracy: absolute, in units of lr)
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=5e-4 * 1e-5)
    for a, b in zip(outs[0][1:3], outs[1][1:3]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-12)


@gpu
@pytest.mark.parametrize("batch", [200, 20000])
def test_training_with_the_norm_rider_is_deterministic_and_matches_the_norm_kernel(batch, monkeypatch):
    torch.manual_seed(0)
    theta = torch.randn(30000, 10)
    x = theta + 0.3 * torch.randn(30000, 10)
    runs = []
    for flag in ("1", "1", "0"):
        monkeypatch.setenv("SBI_AMD_NORM_RIDER", flag)
        torch.manual_seed(1)
        est = build_nsf(theta, x).cuda()
        st = FusedTrainStep(est)
        tb, xb = theta[:batch].cuda(), x[:batch].cuda()
        losses = torch.stack([st.step(tb, xb).mean() for _ in range(6)])
        norms = st.grad_norm().clone()
        runs.append((est.net.flat_params.detach().clone(), losses, norms))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])      # run to run: bit-identical
    assert torch.allclose(runs[0][0], runs[2][0], rtol=1e-5, atol=1e-7)                     # rider vs norm kernel
    assert torch.allclose(runs[0][1], runs[2][1], rtol=1e-5, atol=1e-6)
    assert torch.allclose(runs[0][2], runs[2][2], rtol=1e-6)


@gpu
def test_alternating_batch_sizes_keep_one_table_per_image(monkeypatch):
    """Steps on both sides of the kernel families' threshold (200 rows: cooperative image, 20 000: throughput image) in
    turn: two tables live side by side (no rebuild per step), every step re-packs the image it just made stale for the
    NEXT reader through `packed_weights`, and the run is bit-identical to the one that re-packs with the pack kernels."""
    torch.manual_seed(0)
    theta = torch.randn(30000, 10)
    x = theta + 0.3 * torch.randn(30000, 10)
    runs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SBI_AMD_FUSED_TAIL", flag)
        torch.manual_seed(1)
        est = build_nsf(theta, x).cuda()
        st = FusedTrainStep(est)
        losses = []
        for i in range(6):
            b = 200 if i % 2 == 0 else 20000
            losses.append(st.step(theta[:b].cuda(), x[:b].cuda()).mean())
        if flag == "1":
            assert sorted(k[0] for k in st._step_maps) == [1, 2]
        runs[flag] = (est.net.flat_params.detach().clone(), torch.stack(losses))
    assert torch.equal(runs["1"][0], runs["0"][0]) and torch.equal(runs["1"][1], runs["0"][1])


@gpu
def test_a_gradient_changed_by_hand_gets_its_norm_from_the_gradient_itself():
    torch.manual_seed(0)
    theta = torch.randn(4000, 10)
    x = theta + 0.3 * torch.randn(4000, 10)
    est = build_nsf(theta, x).cuda()
    st = FusedTrainStep(est)
    tb, xb = theta[:512].cuda(), x[:512].cuda()
    st.loss_and_grad(tb, xb)
    want = float(st.grad.double().norm())
    st.apply()
    assert abs(float(st.grad_norm()) - want) <= 1e-5 * want            # rider: the pass's own partial sums
    st.loss_and_grad(tb, xb)
    st.grad.mul_(3.0)
    st.grad_modified()
    want3 = float(st.grad.double().norm())
    st.apply()
    assert abs(float(st.grad_norm()) - want3) <= 1e-5 * want3          # norm kernel over the modified gradient
