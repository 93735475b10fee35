"""ctypes binding of the C ABI declared in include/sbi_amd_nsf.h.

There is no CPU fallback: if the shared library cannot be loaded, or a call is
made with tensors that are not on a ROCm device, the call raises.
"""

from __future__ import annotations

import ctypes
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p
from typing import Optional

import torch

from sbi_amd import _build

_LIB: Optional[ctypes.CDLL] = None
ABI_VERSION = 114    # must equal sbi_amd_nsf_abi_version() (csrc/nsf_plan.cpp) and SBI_AMD_NSF_ABI_VERSION (include/)

E_UNSUPPORTED, E_BADARG, E_LDS = -1, -2, -3
_ERRORS = {
    E_UNSUPPORTED: "configuration not supported by the HIP kernels "
    "(need 1<=D<=64, hidden_features<=64 (<=128 for theta-dim 2..16 with x-dim<=32), num_bins in {4,5,8,10,16}, num_transforms<=16, num_blocks<=4)",
    E_BADARG: "bad argument",
    E_LDS: "configuration needs more than 160 KiB of LDS per workgroup (one transform's weight image plus the "
    "kernel's tiles must fit: e.g. with hidden_features=50, 10 bins, 2 blocks training reaches x-dim 94 at theta-dim "
    "10, x-dim 54 at theta-dim 20, theta-dim 24 with x-dim 32; hidden_features=32 reaches theta-dim 32 with x-dim 48; "
    "smaller hidden_features / theta-dim / x-dim / num_bins fit more, an embedding_net shrinks x-dim)",
}


class NSFConfigC(Structure):
    """Mirror of ``struct sbi_amd_nsf_config``."""

    _fields_ = [
        ("D", c_int32), ("C", c_int32), ("H", c_int32), ("K", c_int32), ("T", c_int32), ("NB", c_int32),
        ("tail_bound", c_float), ("min_bin_width", c_float), ("min_bin_height", c_float),
        ("min_derivative", c_float), ("lu_eps", c_float), ("ctx_layers", c_int32),
    ]


class MAFConfigC(Structure):
    """Mirror of ``struct sbi_amd_maf_config`` (include/sbi_amd_maf.h)."""

    _fields_ = [
        ("D", c_int32), ("C", c_int32), ("H", c_int32), ("K", c_int32), ("T", c_int32), ("NB", c_int32),
        ("tail_bound", c_float), ("min_bin_width", c_float), ("min_bin_height", c_float),
        ("min_derivative", c_float), ("scale_by_sqrt_hidden", c_int32), ("variant", c_int32),
    ]


class FMPEConfigC(Structure):
    """Mirror of ``struct sbi_amd_fmpe_config`` (include/sbi_amd_fmpe.h)."""

    _fields_ = [
        ("D", c_int32), ("C", c_int32), ("H", c_int32), ("L", c_int32), ("E", c_int32),
        ("max_freq", c_float), ("noise_scale", c_float), ("ln_eps", c_float),
    ]


_SIGNATURES = {
    "sbi_amd_nsf_param_count": (c_int64, [POINTER(NSFConfigC)]),
    "sbi_amd_nsf_layer_offset": (c_int64, [POINTER(NSFConfigC), c_int32]),
    "sbi_amd_nsf_lu_offset": (c_int64, [POINTER(NSFConfigC), c_int32]),
    "sbi_amd_nsf_packed_floats": (c_int64, [POINTER(NSFConfigC)]),
    "sbi_amd_nsf_pack": (c_int, [POINTER(NSFConfigC), c_void_p, c_void_p, c_void_p]),
    "sbi_amd_nsf_image_kind": (c_int, [POINTER(NSFConfigC), c_int64, c_int32]),
    "sbi_amd_nsf_set_coop_max_rows": (c_int64, [c_int64]),
    "sbi_amd_nsf_coop_selfcheck": (c_int, [POINTER(NSFConfigC)]),
    "sbi_amd_nsf_pack_images": (c_int, [POINTER(NSFConfigC), c_void_p, c_void_p, c_int32, c_void_p]),
    "sbi_amd_nsf_log_prob": (
        c_int,
        [POINTER(NSFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
         c_void_p],
    ),
    "sbi_amd_nsf_sample": (
        c_int,
        [POINTER(NSFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
         c_void_p],
    ),
    "sbi_amd_nsf_train_workspace_floats": (c_int64, [POINTER(NSFConfigC), c_int64]),
    "sbi_amd_nsf_loss_fwd_bwd": (
        c_int,
        [POINTER(NSFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
         c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_nsf_train_forward": (
        c_int,
        [POINTER(NSFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
         c_void_p],
    ),
    "sbi_amd_nsf_train_backward": (
        c_int,
        [POINTER(NSFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_float,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_adam_clip_step": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_float, c_float,
         c_void_p, c_void_p],
    ),
    "sbi_amd_nsf_train_sqnorm_parts": (c_void_p, [POINTER(NSFConfigC), c_int64, c_void_p, POINTER(c_int64)]),
    "sbi_amd_adam_clip_step_parts": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_float, c_float,
         c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "sbi_amd_nsf_step_map_ints": (c_int64, [POINTER(NSFConfigC)]),
    "sbi_amd_nsf_step_map_workspace_floats": (c_int64, [POINTER(NSFConfigC)]),
    "sbi_amd_nsf_build_step_map": (
        c_int, [POINTER(NSFConfigC), c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sbi_amd_nsf_table_pack": (c_int, [POINTER(NSFConfigC), c_void_p, c_void_p, c_void_p, c_void_p]),
    "sbi_amd_nsf_release_step_map": (c_int, [c_void_p]),
    "sbi_amd_rccl_unique_id_bytes": (c_int32, []),
    "sbi_amd_rccl_unique_id": (c_int, [c_void_p]),
    "sbi_amd_rccl_comm_init": (c_int, [POINTER(c_void_p), c_int32, c_int32, c_void_p]),
    "sbi_amd_rccl_comm_destroy": (c_int, [c_void_p]),
    "sbi_amd_allreduce_flat": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sbi_amd_mcmc_slice_run": (
        c_int,
        [POINTER(NSFConfigC), c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_uint64, c_int32, c_int32, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_atomic_atoms": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sbi_amd_atomic_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                       c_void_p]),
    "sbi_amd_accept_compact_scan_words": (c_int64, [c_int64, c_int32]),
    "sbi_amd_accept_compact": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p,
         c_void_p, ctypes.c_uint32, c_void_p],
    ),
    "sbi_amd_shuffled_gather": (
        c_int,
        [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int64, c_uint64, c_int64, c_int64, c_void_p, c_void_p,
         c_void_p, c_void_p],
    ),
    "sbi_amd_rq_spline": (
        c_int,
        [c_int32, c_int32, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
         c_void_p],
    ),
    "sbi_amd_mcmc_slice_tick": (
        c_int,
        [c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_uint64, c_int32, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p],
    ),
    "sbi_amd_mcmc_to_constrained": (
        c_int,
        [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_fmpe_param_count": (c_int64, [POINTER(FMPEConfigC)]),
    "sbi_amd_fmpe_param_offset": (c_int64, [POINTER(FMPEConfigC), c_int32, c_int32]),
    "sbi_amd_fmpe_packed_floats": (c_int64, [POINTER(FMPEConfigC)]),
    "sbi_amd_fmpe_pack": (c_int, [POINTER(FMPEConfigC), c_void_p, c_void_p, c_void_p]),
    "sbi_amd_fmpe_velocity": (
        c_int,
        [POINTER(FMPEConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_void_p],
    ),
    "sbi_amd_fmpe_velocity_div": (
        c_int,
        [POINTER(FMPEConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_fmpe_loss": (
        c_int,
        [POINTER(FMPEConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
         c_void_p, c_void_p],
    ),
    "sbi_amd_fmpe_train_workspace_floats": (c_int64, [POINTER(FMPEConfigC), c_int64]),
    "sbi_amd_fmpe_loss_fwd_bwd": (
        c_int,
        [POINTER(FMPEConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
         c_int64, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_dopri5_init": (c_int, [c_void_p, c_double, c_double, c_double, c_double, c_double, c_void_p]),
    "sbi_amd_dopri5_stage": (c_int, [c_void_p, POINTER(c_void_p), c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "sbi_amd_dopri5_finish": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_int64, c_void_p]),
    "sbi_amd_maf_param_count": (c_int64, [POINTER(MAFConfigC)]),
    "sbi_amd_maf_packed_floats": (c_int64, [POINTER(MAFConfigC)]),
    "sbi_amd_maf_param_offset": (c_int64, [POINTER(MAFConfigC), c_int32, c_int32, c_int32]),
    "sbi_amd_maf_pack": (c_int, [POINTER(MAFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sbi_amd_maf_log_prob": (
        c_int,
        [POINTER(MAFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_maf_sample": (
        c_int,
        [POINTER(MAFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_maf_train_workspace_floats": (c_int64, [POINTER(MAFConfigC), c_int64]),
    "sbi_amd_maf_loss_fwd_bwd": (
        c_int,
        [POINTER(MAFConfigC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_float,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "sbi_amd_nsf_plan_waves": (c_int, [POINTER(NSFConfigC), c_int64, c_int32]),
    "sbi_amd_nsf_abi_version": (c_int, []),
    "sbi_amd_nsf_arch": (c_char_p, []),
}


def exported_symbols():
    return list(_SIGNATURES)


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load libsbi_amd_nsf.so (building it in-tree with hipcc if needed)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB_PATH
    import os

    override = os.environ.get("SBI_AMD_LIB")
    if override:      # developer aid (the -DNSF_DEBUG build of tools/timeline.py): never silent
        import sys

        print(f"sbi_amd: WARNING: SBI_AMD_LIB={override}: loading a non-default kernel library "
              "(debug / timing build: results may be INVALID)", file=sys.stderr)
        path = _build.Path(override)
        build_if_missing = False
        if not path.exists():
            raise RuntimeError(f"SBI_AMD_LIB={override} does not exist")
    if override:
        pass
    elif not path.exists() and not build_if_missing:
        raise RuntimeError(f"{path} is missing; run `python -c 'import __graft_entry__ as g; g.build()'`")
    if build_if_missing:
        try:
            _build.build()      # no-op when the library matches the sources on disk (content hash, file-locked)
        except _build.HipccMissing as e:
            # a machine with a prebuilt library but no hipcc must stay usable -- but only with a library that was built
            # from THESE sources: kernel / packed-image layouts change without an ABI-version bump, so the stored
            # content hash is compared whenever it exists, and a library without one is loaded with a loud warning
            if not path.exists():
                raise
            import warnings

            if _build.HASH_PATH.exists():
                if _build.HASH_PATH.read_text().strip() != _build.source_hash():
                    raise RuntimeError(f"{path} was built from different sources than the ones on disk and cannot be "
                                       f"rebuilt here ({e}); refusing to load a stale kernel library") from e
            else:
                warnings.warn(f"sbi_amd: {path.name} has no source hash next to it and cannot be checked against the "
                              f"sources ({e}); loading it as is (only the ABI version is verified)", stacklevel=2)
    elif not override and _build.needs_build():
        raise RuntimeError(f"{path} is stale (built from different sources); rebuild it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(str(path))
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.sbi_amd_nsf_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {got}, this binding expects {ABI_VERSION} (stale library?)")
    _LIB = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"sbi_amd: {what}: {_ERRORS.get(rc, rc)}")
    raise RuntimeError(f"sbi_amd: {what}: HIP error {rc}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_device(*tensors: torch.Tensor) -> torch.device:
    """All tensors must live on one ROCm device, fp32, contiguous."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "sbi_amd: the NSF hot path runs only on a ROCm device (MI355X); got a "
                f"{t.device} tensor. There is deliberately no CPU fallback."
            )
        if t.dtype != torch.float32:
            raise TypeError(f"sbi_amd: expected float32, got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError("sbi_amd: expected contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"sbi_amd: tensors on different devices ({dev} vs {t.device})")
    assert dev is not None
    return dev


def current_stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream
