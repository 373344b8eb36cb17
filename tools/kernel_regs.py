#!/usr/bin/env python3
"""List VGPR / AGPR / SGPR / LDS / scratch per kernel from a hipcc --save-temps .s file (tuning aid, no GPU needed).
usage: hipcc -O3 --offload-arch=gfx950 --save-temps -c x.hip -o /tmp/x.o ; python tools/kernel_regs.py x-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    try: name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    except Exception: pass
    ag = re.match(r"\s*(\d+)", blk).group(1)
    print("%-48s vgpr %4s agpr %3s sgpr %3s lds %6s scratch %5s" % (name, g("vgpr_count"), ag, g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
