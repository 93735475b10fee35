#!/usr/bin/env python
"""Developer aid: build a VARIANT of the kernel library for A/B runs on the GPU box.
usage: python tools/r6/variant.py <name> "<extra hipcc flags>" unit [unit ...]
  -> sbi_amd/libsbi_amd_nsf_<name>.so  (the named translation units recompiled with the flags, every other object
     taken from the standard build); select it with SBI_AMD_LIB=<path>."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd import _build
name, extra, units = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
srcs = {s.rsplit(".", 1)[0]: s for s in _build.SOURCES}
odir = _build.CSRC / ("build_" + name); odir.mkdir(exist_ok=True)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *extra]
def one(n):
    r = subprocess.run([_build.hipcc_path(), *flags, "-c", str(_build.CSRC / srcs[n]), "-o", str(odir / (n + ".o"))], capture_output=True, text=True)
    if r.returncode: print(r.stderr[-4000:]); raise SystemExit(1)
with ThreadPoolExecutor(8) as ex: list(ex.map(one, units))
objs = [str((odir if s.rsplit(".", 1)[0] in units else _build.CSRC / "build") / (s.rsplit(".", 1)[0] + ".o")) for s in _build.SOURCES]
out = _build.LIB_PATH.with_name(f"libsbi_amd_nsf_{name}.so")
subprocess.check_call([_build.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(out)])
print(out)
