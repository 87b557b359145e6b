#!/usr/bin/env python3
"""Per-kernel register / spill / scratch figures from the AMDGPU metadata of a `hipcc --save-temps` device assembly file.
usage: kernel_regs.py <file.s> [name substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt not in name:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    print("%-110s vgpr %4s spill %3s scratch %4s lds %6s sgpr %3s" % (name[:110], g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                                                                      g("group_segment_fixed_size"), g("sgpr_count")))
