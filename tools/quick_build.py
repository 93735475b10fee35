#!/usr/bin/env python
"""Developer aid: recompile only the named translation units (in parallel) and relink libsbi_amd_nsf.so.
usage: python tools/quick_build.py nsf_coop nsf_coop_k4 ...   (no arguments: every nsf_coop* unit)"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sbi_amd import _build   # noqa: E402

names = sys.argv[1:] or [s.rsplit(".", 1)[0] for s in _build.SOURCES if s.startswith("nsf_coop")]
srcs = {s.rsplit(".", 1)[0]: s for s in _build.SOURCES}
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def one(n):
    r = subprocess.run([_build.hipcc_path(), *flags, "-c", str(_build.CSRC / srcs[n]), "-o",
                        str(_build.CSRC / "build" / (n + ".o"))], capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-4000:])
        raise SystemExit(1)


with ThreadPoolExecutor(12) as ex:
    list(ex.map(one, names))
objs = [str(_build.CSRC / "build" / (s.rsplit(".", 1)[0] + ".o")) for s in _build.SOURCES]
subprocess.check_call([_build.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(_build.LIB_PATH)])
_build.HASH_PATH.write_text(_build.source_hash() + "\n")
print("rebuilt", names)
