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
    torch.cuda.synchronize()


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
