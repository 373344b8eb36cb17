#!/usr/bin/env python3
"""Mean PMC counter values per launch of the kernels whose name contains SUBSTR, grouped by grid:
    pmc_kernel.py COUNTER_DIR SUBSTR      (COUNTER_DIR: rocprofv3 --kernel-trace --pmc ... --output-format csv -d COUNTER_DIR)"""
import collections
import csv
import glob
import sys

d, sub = sys.argv[1], sys.argv[2]
files = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    n = max(len(v) for v in cs.values())
    print("%s grid %s (%d launches)" % (key[0], key[1], n))
    for c, v in sorted(cs.items()):
        print("    %-28s %14.0f" % (c, sum(v) / len(v)))
