#!/usr/bin/env python3
"""Scan a hipcc --save-temps gfx950 .s file for waits that make a wave sit out the acknowledgement of its own global STORES: on gfx9 loads and
stores share vmcnt and return in order, so an `s_waitcnt vmcnt(N)` for a LOAD issued after stores also waits for those stores (an HBM write round
trip).  Per kernel: the number of vmcnt waits that have at least one store among the operations they wait for, and the stores so exposed.
usage: hipcc -O3 --offload-arch=gfx950 --save-temps -c x.hip ; python tools/isa_store_waits.py x-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if flt not in name:
        continue
    queue = []           # outstanding vector memory operations in issue order: 'L' / 'S'
    waits = exposed = 0
    lines = m.group(2).split("\n")
    first = None
    for i, l in enumerate(lines):
        l = l.strip()
        if re.match(r"(global|buffer|flat|scratch)_load", l): queue.append("L")
        elif re.match(r"(global|buffer|flat|scratch)_store", l) or re.match(r"(global|buffer|flat)_atomic", l): queue.append("S")
        else:
            w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if w:
                n = int(w.group(1))
                done = queue[:len(queue) - n] if n else queue
                k = done.count("S")
                if k and "L" in queue:           # a wait with stores among what it waits for (a straight-line estimate: branches are ignored)
                    waits += 1; exposed += k
                    if first is None: first = i
                queue = queue[len(queue) - n:] if n else []
    if waits:
        print("%-56s %3d waits behind %4d stores (first at line %d of the kernel)" % (name[:56], waits, exposed, first))
