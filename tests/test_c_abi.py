"""The C-ABI library loads and exports every symbol include/*.h declares (no
compute calls here: those need a GPU).  Host-only entry points are exercised."""

import ctypes
import os
import re

import pytest

from sbi_amd import _build, _lib
from sbi_amd.neural_nets.estimators.nsf_flow import NSFHyper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for header in ("sbi_amd_nsf.h", "sbi_amd_fmpe.h", "sbi_amd_maf.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(sbi_amd_\w+)\s*\(", text))
    return sorted(syms)


def test_library_builds_and_exports_every_declared_symbol():
    _build.build()
    lib = ctypes.CDLL(str(_build.LIB_PATH))
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/*.h but not exported"
    assert set(syms) == set(_lib.exported_symbols()), "ctypes binding and header disagree"


def test_version_arch_and_layout_helpers():
    lib = _lib.load()
    assert lib.sbi_amd_nsf_abi_version() >= 100
    assert lib.sbi_amd_nsf_arch() == b"gfx950"
    for kw in [dict(D=10, C=10), dict(D=2, C=2), dict(D=3, C=5, hidden_features=32, num_transforms=3, num_bins=8,
                                                       num_blocks=1), dict(D=7, C=4, num_bins=4)]:
        h = NSFHyper(**kw)
        cfg = h.c_config()
        assert lib.sbi_amd_nsf_param_count(cfg) == h.param_count()
        off = 0
        for t in range(h.num_transforms):
            assert lib.sbi_amd_nsf_layer_offset(cfg, t) == off
            n_layer = sum(int(__import__("numpy").prod(s)) for _, s in h.layer_entries(t))
            assert lib.sbi_amd_nsf_lu_offset(cfg, t) == off + n_layer
            off += n_layer + h.D * (h.D - 1) + 2 * h.D
        assert lib.sbi_amd_nsf_packed_floats(cfg) > h.param_count()
    assert NSFHyper(D=10, C=10).param_count() == 98025


def test_unsupported_configs_are_refused_not_degraded():
    lib = _lib.load()
    assert lib.sbi_amd_nsf_param_count(NSFHyper(D=10, C=10, hidden_features=129).c_config()) == _lib.E_UNSUPPORTED
    # hidden 65 ... 128: the wide cooperative kernels (every batch size reads the cooperative image); shapes those
    # kernels do not take (theta-dim 1, theta-dim > 16, x-dim > 32) have no kernel at that width and say so
    wide = NSFHyper(D=10, C=10, hidden_features=128)
    assert lib.sbi_amd_nsf_param_count(wide.c_config()) == wide.param_count() > 0
    assert lib.sbi_amd_nsf_image_kind(wide.c_config(), 65536, 0) == 1
    assert lib.sbi_amd_nsf_packed_floats(wide.c_config()) > 0
    assert lib.sbi_amd_nsf_image_kind(NSFHyper(D=5, C=64, hidden_features=100).c_config(), 100, 0) == 1
    for kw in (dict(D=1, C=3), dict(D=20, C=5), dict(D=5, C=65)):
        assert lib.sbi_amd_nsf_image_kind(NSFHyper(hidden_features=100, **kw).c_config(), 100, 0) == _lib.E_UNSUPPORTED
    assert lib.sbi_amd_nsf_param_count(NSFHyper(D=10, C=10, num_bins=7).c_config()) == _lib.E_UNSUPPORTED
    assert lib.sbi_amd_nsf_param_count(NSFHyper(D=0, C=10).c_config()) == _lib.E_BADARG
    assert lib.sbi_amd_nsf_param_count(NSFHyper(D=1, C=3).c_config()) == NSFHyper(D=1, C=3).param_count() == 21145
    # null pointers never reach a launch
    cfg = NSFHyper(D=10, C=10).c_config()
    assert lib.sbi_amd_nsf_log_prob(cfg, None, None, None, None, 4, 4, None, None, None) == _lib.E_BADARG
    assert lib.sbi_amd_adam_clip_step(None, None, None, None, 4, 1, 1e-3, 0.9, 0.999, 1e-8, 5.0, None, None) == \
        _lib.E_BADARG
    with pytest.raises(RuntimeError, match="unsupported|not supported"):
        _lib.check(_lib.E_UNSUPPORTED, "x")


def test_throughput_workgroup_size_minimises_rounds_over_the_cus():
    """csrc/nsf_plan.cpp::nsf_plan_for_rows: 16 rows per wave, 1 / 2 / 4 / 8 waves per workgroup, one workgroup per CU at a
    time => ceil(workgroups / 256) rounds of about equal length; the size is chosen to minimise the rounds (x 1.2 for eight
    waves), smaller workgroups among equals.  12 288 rows used to run as 384 two-wave workgroups (two rounds)."""
    lib = _lib.load()
    c = NSFHyper(D=10, C=10).c_config()
    got = {n: lib.sbi_amd_nsf_plan_waves(c, n, 0) for n in (1, 200, 4096, 4097, 8192, 8193, 12288, 16384, 16385, 24576,
                                                            32768, 40000, 65536, 10**6)}
    assert got == {1: 1, 200: 1, 4096: 1, 4097: 2, 8192: 2, 8193: 4, 12288: 4, 16384: 4, 16385: 8, 24576: 8, 32768: 8,
                   40000: 8, 65536: 8, 10**6: 8}, got
    # the sampling direction takes twelve-wave workgroups once they fill the chip
    assert lib.sbi_amd_nsf_plan_waves(c, 10**6, 1) == 12 and lib.sbi_amd_nsf_plan_waves(c, 12288, 1) == 4
    # a net wider than 64 has no throughput kernel at all
    assert lib.sbi_amd_nsf_plan_waves(NSFHyper(D=10, C=10, hidden_features=100).c_config(), 4096, 0) == _lib.E_LDS


def test_training_envelope_and_refusals_are_host_side_decisions():
    """`sbi_amd_nsf_train_workspace_floats` answers on the host (no device call) whether a shape trains: the
    wave-specialised backward kernel's shapes, the generic training pass beyond them (theta-dim > 15, 3-4 blocks,
    wide x), E_LDS once a transform's weight image plus the kernel's tiles no longer fit 160 KiB."""
    lib = _lib.load()

    def ws(**kw):
        return lib.sbi_amd_nsf_train_workspace_floats(NSFHyper(**kw).c_config(), 4096)

    # the benchmark shape and the generic-pass shapes of tests/test_nsf_train_gpu.py
    for kw in (dict(D=10, C=10), dict(D=1, C=3), dict(D=15, C=20, num_transforms=3),
               dict(D=10, C=10, hidden_features=64, num_transforms=2),
               dict(D=10, C=10, num_blocks=3, num_transforms=2), dict(D=20, C=10, num_transforms=2),
               dict(D=32, C=6, num_transforms=3, num_bins=8), dict(D=16, C=40, num_transforms=2),
               dict(D=12, C=70, num_transforms=2, hidden_features=48),
               dict(D=6, C=60, num_transforms=2, num_blocks=4, hidden_features=32)):
        assert ws(**kw) > 0, kw
    # the envelope DESIGN.md section 1 quotes for sbi's default hyper-parameters (hidden 50, 10 bins, 2 blocks)
    for D, cmax in ((2, 127), (5, 110), (10, 94), (16, 66), (20, 54), (24, 32)):
        assert ws(D=D, C=cmax) > 0, (D, cmax)
        assert ws(D=D, C=cmax + 1) == _lib.E_LDS, (D, cmax)
    assert ws(D=32, C=4) == _lib.E_LDS
    # a larger workspace for a larger batch, and the generic pass needs more of it than the fused one
    small = lib.sbi_amd_nsf_train_workspace_floats(NSFHyper(D=10, C=10).c_config(), 1024)
    big = lib.sbi_amd_nsf_train_workspace_floats(NSFHyper(D=10, C=10).c_config(), 65536)
    assert 0 < small < big
    # hidden 65 ... 128 trains on the wide cooperative kernels at every batch size; beyond, and for the shapes those
    # kernels do not take, the refusal is a host-side answer as well
    assert lib.sbi_amd_nsf_train_workspace_floats(NSFHyper(D=10, C=10, hidden_features=128).c_config(), 1024) > 0
    assert lib.sbi_amd_nsf_train_workspace_floats(NSFHyper(D=10, C=10, hidden_features=100).c_config(), 65536) > 0
    assert lib.sbi_amd_nsf_train_workspace_floats(NSFHyper(D=10, C=10, hidden_features=129).c_config(), 1024) == \
        _lib.E_UNSUPPORTED
    assert lib.sbi_amd_nsf_train_workspace_floats(NSFHyper(D=20, C=10, hidden_features=100).c_config(), 1024) < 0
