"""Misuse of the C ABI that round 5 refuses instead of trusting:
* `sbi_amd_nsf_table_pack` with a table this process did not build for the same configuration;
* a training pass whose backward half would take a different kernel family than its forward half took (the
  process-wide threshold moved in between)."""
import pytest
import torch

from sbi_amd import _lib
from sbi_amd.neural_nets.estimators.nsf_flow import packed_weights, train_backward, train_forward, train_workspace
from tests.helpers import matched_pair

pytestmark = pytest.mark.gpu


def test_table_pack_refuses_tables_it_did_not_build():
    lib = _lib.load()
    _, est, _, _ = matched_pair(D=4, C=3, hidden_features=32, num_transforms=2)
    _, other, _, _ = matched_pair(D=5, C=3, hidden_features=32, num_transforms=2)
    cfg, cfg_other = est.net.hyper.c_config(), other.net.hyper.c_config()
    packed = packed_weights(est.net, rows=None)
    stream = _lib.current_stream(packed.device)
    table = torch.zeros(int(lib.sbi_amd_nsf_step_map_ints(cfg)), dtype=torch.int32, device="cuda")
    scratch = torch.empty(int(lib.sbi_amd_nsf_step_map_workspace_floats(cfg)), device="cuda")
    flat = est.net.flat_params.data
    # never built: refused
    assert lib.sbi_amd_nsf_table_pack(cfg, _lib.ptr(flat), _lib.ptr(packed), _lib.ptr(table), stream) == _lib.E_BADARG
    assert lib.sbi_amd_nsf_build_step_map(cfg, 1, _lib.ptr(flat), _lib.ptr(packed), _lib.ptr(table), _lib.ptr(scratch),
                                          stream) == 0
    assert lib.sbi_amd_nsf_table_pack(cfg, _lib.ptr(flat), _lib.ptr(packed), _lib.ptr(table), stream) == 0
    # built for another configuration: refused
    assert lib.sbi_amd_nsf_table_pack(cfg_other, _lib.ptr(other.net.flat_params.data), _lib.ptr(packed), _lib.ptr(table),
                                      stream) == _lib.E_BADARG
    # released (the owner is about to free the buffer): refused again, and building 2 000 other tables in between
    # never evicts a live one (the registry used to clear itself at 1 024 entries)
    assert lib.sbi_amd_nsf_release_step_map(_lib.ptr(table)) == 0
    assert lib.sbi_amd_nsf_table_pack(cfg, _lib.ptr(flat), _lib.ptr(packed), _lib.ptr(table), stream) == _lib.E_BADARG
    assert lib.sbi_amd_nsf_build_step_map(cfg, 1, _lib.ptr(flat), _lib.ptr(packed), _lib.ptr(table), _lib.ptr(scratch),
                                          stream) == 0
    torch.cuda.synchronize()


def test_fused_step_survives_a_released_table():
    """A table the library no longer knows must not raise AFTER the optimizer step: the stepper falls back to the full
    pack and rebuilds the table."""
    from sbi_amd.inference.trainers.fused import FusedTrainStep

    lib = _lib.load()
    _, est, theta, x = matched_pair(D=10, C=10)
    th, xx = theta[:512].cuda().contiguous(), x[:512].cuda().contiguous()
    st, ref = FusedTrainStep(est, distributed=False), None
    st.step(th, xx)
    st.step(th, xx)
    maps = [v for v in st.__dict__.get("_step_maps", {}).values() if isinstance(v, torch.Tensor)]
    assert maps, "the stepper did not build a re-pack table"
    for mp in maps:
        lib.sbi_amd_nsf_release_step_map(mp.data_ptr())
    st.step(th, xx)                     # table refused -> full re-pack on demand, no exception
    losses = st.step(th, xx)            # table rebuilt
    # the same four steps on a twin that never lost its table: identical parameters
    _, twin, _, _ = matched_pair(D=10, C=10)
    st2 = FusedTrainStep(twin, distributed=False)
    for _ in range(4):
        losses2 = st2.step(th, xx)
    torch.cuda.synchronize()
    assert torch.equal(est.net.flat_params.data, twin.net.flat_params.data) and torch.equal(losses, losses2)


def test_backward_refuses_a_stash_of_the_other_kernel_family():
    lib = _lib.load()
    _, est, theta, x = matched_pair(D=10, C=10)
    n = 512
    th, xx = theta[:n].cuda().contiguous(), x[:n].cuda().contiguous()
    grad = torch.empty_like(est.net.flat_params.data)
    w = torch.full((n,), 1.0 / n, device="cuda")
    prev = lib.sbi_amd_nsf_set_coop_max_rows(12288)
    try:
        ws = train_workspace(est.net, n, "cuda")
        # the workspace must be large enough for either family
        lib.sbi_amd_nsf_set_coop_max_rows(0)
        ws = train_workspace(est.net, n, "cuda", ws)
        lib.sbi_amd_nsf_set_coop_max_rows(12288)
        train_forward(est.net, th, xx, ws)                       # cooperative forward lays the stash out
        lib.sbi_amd_nsf_set_coop_max_rows(0)                     # ... the threshold moves ...
        est.net.__dict__.pop("_packed_cache", None)
        with pytest.raises(RuntimeError, match="bad argument|BADARG|-2"):
            train_backward(est.net, xx, n, w, grad, ws)          # ... the throughput backward is refused
        lib.sbi_amd_nsf_set_coop_max_rows(12288)
        est.net.__dict__.pop("_packed_cache", None)
        train_backward(est.net, xx, n, w, grad, ws)              # the matching family still runs
        torch.cuda.synchronize()
        assert torch.isfinite(grad).all()
    finally:
        lib.sbi_amd_nsf_set_coop_max_rows(prev)
