"""Host side of the cooperative small-batch kernels (csrc/nsf_coop*.{h,cpp,hip}) without a GPU: which calls they take,
the packed-buffer bookkeeping, the workspace layout, and the self-check that holds the kernels' address ARITHMETIC
(they read ~100 scalars, not the plan tables) to the tables the pack / reduce kernels use."""
import pytest

from sbi_amd import _lib
from sbi_amd.neural_nets.estimators.nsf_flow import NSFHyper


def _cfg(**kw):
    base = dict(D=10, C=10, hidden_features=50, num_bins=10, num_transforms=5, num_blocks=2)
    base.update(kw)
    return NSFHyper(**base).c_config()


SHAPES = [dict(), dict(D=2, C=2), dict(D=16, C=32), dict(D=5, C=3, hidden_features=64, num_blocks=4, num_bins=16),
          dict(D=7, C=17, hidden_features=33, num_bins=4, num_transforms=3, num_blocks=1), dict(D=15, C=20, num_bins=8),
          # hidden 65 ... 128: the wide kernels (eight hidden m-tiles, eight K-quads, x-dim up to 64, packed LU inverses)
          dict(hidden_features=100), dict(hidden_features=128, num_blocks=4), dict(D=16, C=64, hidden_features=65, num_bins=16),
          dict(D=2, C=33, hidden_features=96, num_bins=4, num_transforms=2, num_blocks=1)]


@pytest.mark.parametrize("kw", SHAPES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()) or "default")
def test_kernel_arithmetic_matches_the_plan_tables(kw):
    assert _lib.load().sbi_amd_nsf_coop_selfcheck(_cfg(**kw)) == 0


def test_shapes_outside_the_cooperative_family_keep_the_throughput_kernels():
    lib = _lib.load()
    for kw in (dict(D=1, C=3), dict(D=17, C=3), dict(D=4, C=33)):
        c = _cfg(**kw)
        assert lib.sbi_amd_nsf_coop_selfcheck(c) == -1
        assert lib.sbi_amd_nsf_image_kind(c, 200, 1) == 0 and lib.sbi_amd_nsf_image_kind(c, 200, 0) == 0


def test_row_threshold_routes_calls_and_never_changes_the_packed_size():
    lib, c = _lib.load(), _cfg()
    size = lib.sbi_amd_nsf_packed_floats(c)
    prev = lib.sbi_amd_nsf_set_coop_max_rows(12288)
    try:
        assert [lib.sbi_amd_nsf_image_kind(c, n, 1) for n in (1, 200, 12288, 12289, 65536)] == [1, 1, 1, 0, 0]
        ws_small = lib.sbi_amd_nsf_train_workspace_floats(c, 200)
        assert lib.sbi_amd_nsf_set_coop_max_rows(0) == 12288
        assert lib.sbi_amd_nsf_image_kind(c, 200, 1) == 0
        assert lib.sbi_amd_nsf_packed_floats(c) == size          # both images stay in the buffer
        assert lib.sbi_amd_nsf_train_workspace_floats(c, 200) != ws_small   # the workspace belongs to the family
    finally:
        lib.sbi_amd_nsf_set_coop_max_rows(prev)
    assert size > 5 * 20000 * 2                                   # throughput image + forward and transposed coop image


def test_wide_nets_take_the_cooperative_path_at_every_batch_size():
    lib = _lib.load()
    c = _cfg(hidden_features=100)
    prev = lib.sbi_amd_nsf_set_coop_max_rows(0)          # the switch that turns the narrow cooperative path off ...
    try:
        assert [lib.sbi_amd_nsf_image_kind(c, n, t) for n in (1, 200, 65536, 10**6) for t in (0, 1)] == [1] * 8   # ... is ignored
        assert lib.sbi_amd_nsf_train_workspace_floats(c, 65536) > lib.sbi_amd_nsf_train_workspace_floats(c, 200) > 0
    finally:
        lib.sbi_amd_nsf_set_coop_max_rows(prev)
    # only the cooperative image exists for them; it carries eight m-tiles x eight quads per hidden matrix, both directions
    assert lib.sbi_amd_nsf_packed_floats(c) > 5 * 2 * (2 * 2 * 128 * 128)
    # a 16-bin, theta-dim-16 net has 24 final-layer tiles: the narrow kernels (four per wave) leave it to the throughput
    # kernels, the wide ones loop over them
    assert lib.sbi_amd_nsf_image_kind(_cfg(D=16, C=8, num_bins=16), 200, 0) == 0
    assert lib.sbi_amd_nsf_image_kind(_cfg(D=16, C=8, num_bins=16, hidden_features=80), 200, 0) == 1


def test_default_thresholds_inference_12288_training_8192():
    """Measured cross-overs: log_prob / sampling stay on the cooperative kernels up to 12 288 rows, the
    training pass up to 8 192 (one round of two-tile backward workgroups).  (Fresh process: the setter moves both.)"""
    import subprocess
    import sys

    code = ("from sbi_amd import _lib; from sbi_amd.neural_nets.estimators.nsf_flow import NSFHyper; "
            "lib = _lib.load(); c = NSFHyper(D=10, C=10).c_config(); "
            "print([lib.sbi_amd_nsf_image_kind(c, n, t) for t in (0, 1) for n in (8192, 8193, 12288, 12289)])")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(_lib.__file__).rsplit("/", 2)[0],
                         env={k: v for k, v in __import__("os").environ.items() if k != "SBI_AMD_COOP_MAX_ROWS"})
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "[1, 1, 1, 0, 1, 0, 0, 0]"


def test_workspace_grows_with_rows_and_switches_workgroup_shape():
    lib, c = _lib.load(), _cfg()
    w16, w17, w4096, w4097, w8192 = [lib.sbi_amd_nsf_train_workspace_floats(c, n) for n in (16, 17, 4096, 4097, 8192)]
    assert 0 < w16 < w17 < w4096 and 0 < w4097 < w8192
    # 4 097 rows: two 16-row tiles per workgroup => half as many partial slabs as 4 096 rows (one tile each)
    assert w4097 < w4096
