"""In-tree build of libsbi_amd_nsf.so (hipcc, gfx950 only).

Used by ``__graft_entry__.build()`` and by ``sbi_amd._lib`` when the shared
library is missing.  hipcc cross-compiles without a GPU present.
"""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = Path(__file__).resolve().parent / "libsbi_amd_nsf.so"
SOURCES = ["nsf_plan.cpp", "nsf_flow.hip", "nsf_flow_inv.hip", "nsf_train.hip", "nsf_train_k4.hip", "nsf_train_k5.hip", "nsf_train_k8.hip", "nsf_train_k16.hip", "fmpe.hip", "ode.hip",
           "adam.hip", "spline_abi.hip", "step_tail.hip", "shuffle.hip", "compact.hip", "allreduce.hip", "atomic.hip", "mcmc_slice.hip", "nsf_gtrain.hip", "nsf_coop_plan.cpp", "nsf_coop.hip", "nsf_coop_k4.hip", "nsf_coop_k5.hip", "nsf_coop_k8.hip", "nsf_coop_k16.hip", "maf.hip", "maf_k4.hip", "maf_k5.hip", "maf_k8.hip", "maf_k16.hip"]
HEADERS = ["adam_math.h", "nsf_plan.h", "nsf_plan_layout.h", "nsf_coop_wide_kernel.h", "nsf_device.h", "nsf_flow_kernel.h", "nsf_train_kernel.h", "debug_env.h", "maf_kernel.h", "nsf_gtrain_kernel.h", "nsf_coop.h", "nsf_coop_kernel.h", "nsf_coop_host.h", "../../include/sbi_amd_maf.h",
           "../../include/sbi_amd_nsf.h", "../../include/sbi_amd_fmpe.h"]
HASH_PATH = LIB_PATH.with_suffix(".so.srchash")   # travels with the .so (git-ignored, not gpurun-ignored)
LOCK_PATH = LIB_PATH.with_suffix(".so.lock")


class HipccMissing(RuntimeError):
    """No hipcc on this machine: a prebuilt library can still be loaded, nothing can be rebuilt."""


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise HipccMissing("hipcc not found: the sbi_amd HIP extension cannot be built")


def source_hash() -> str:
    """Content hash of every source / header the library is built from (mtimes do not survive a snapshot copy)."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted([CSRC / s for s in SOURCES] + [CSRC / hd for hd in HEADERS]):
        if f.exists():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    h.update(os.environ.get("SBI_AMD_EXTRA_HIPCC_FLAGS", "").encode())
    return h.hexdigest()


def needs_build() -> bool:
    """True when the library is missing or was built from different sources than the ones on disk."""
    if not LIB_PATH.exists():
        return True
    if not HASH_PATH.exists():
        return True
    return HASH_PATH.read_text().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every translation unit for gfx950 (in parallel) and link the shared library."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl

    # one builder at a time (torchrun ranks, pytest-xdist workers): the others wait, then find it up to date
    with open(LOCK_PATH, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


DEBUG_LIB_PATH = LIB_PATH.with_name("libsbi_amd_nsf_debug.so")


def build_debug(verbose: bool = False) -> Path:
    """Developer aid: the same library with -DNSF_DEBUG (cycle-counter timeline stores, run-time ablation switches) as
    ``libsbi_amd_nsf_debug.so``.  Never loaded by default: ``SBI_AMD_LIB=<path>`` selects it (tools/timeline.py)."""
    return _build_locked(verbose, debug=True)


def _build_locked(verbose: bool, debug: bool = False) -> Path:
    from concurrent.futures import ThreadPoolExecutor

    hipcc = hipcc_path()
    objdir = CSRC / ("build_debug" if debug else "build")
    objdir.mkdir(exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
    flags += os.environ.get("SBI_AMD_EXTRA_HIPCC_FLAGS", "").split()   # experiments only (e.g. -DNSF_PRIO_MODE=1)
    if debug:
        flags.append("-DNSF_DEBUG")

    def compile_one(src: str):
        obj = objdir / (Path(src).stem + ".o")
        cmd = [hipcc, *flags, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{res.stdout}\n{res.stderr}")
        return str(obj)

    srcs = [s for s in SOURCES if (CSRC / s).exists()]
    with ThreadPoolExecutor(max_workers=min(12, len(srcs))) as pool:
        objs = list(pool.map(compile_one, srcs))
    out = DEBUG_LIB_PATH if debug else LIB_PATH
    tmp = out.with_suffix(".so.tmp")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{res.stdout}\n{res.stderr}")
    os.replace(tmp, out)          # atomic: a concurrent dlopen never sees a half-written file
    if not debug:
        HASH_PATH.write_text(source_hash() + "\n")
    return out


if __name__ == "__main__":
    import sys

    print(build_debug(verbose=True) if "--debug" in sys.argv else build(force=True, verbose=True))
