#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection.csv values per kernel (fneus kernels only)."""
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "fneus::" not in name: continue
            short = name.split("fneus::")[1].split("(")[0]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
