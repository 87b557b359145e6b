#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection csv by (kernel, grid) -> counter sums."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:60], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[key].add(r["Dispatch_Id"])
names = sorted({c for v in agg.values() for c in v})
print("kernel,grid,calls," + ",".join(names))
for key in sorted(agg, key=lambda k: -agg[k].get(names[0], 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%s,%s,%d," % (key[0], key[1], len(calls[key])) + ",".join("%.4g" % agg[key].get(n, 0) for n in names))
