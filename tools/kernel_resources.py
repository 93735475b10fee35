#!/usr/bin/env python
"""Print registers / spills / scratch / LDS of every kernel in a hipcc --save-temps .s file (or all of csrc)."""
import re, subprocess, sys, tempfile, pathlib
def table(s_path):
    txt = open(s_path).read()
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        print(f"{name[:58]:58s} agpr {blk.split()[0]:>3} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} "
              f"spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size')}")
if __name__ == "__main__":
    for p in sys.argv[1:]:
        table(p)
