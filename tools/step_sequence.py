#!/usr/bin/env python3
"""List the kernels of ONE train step in launch order from a rocprofv3 --kernel-trace CSV (name, grid, duration).
Usage: step_sequence.py kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rowscale_kernel" in r["Kernel_Name"]]
starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 8]
for r in rows[starts[-2]:starts[-1]]:
    n = r["Kernel_Name"].replace("void ", "").replace("at::native::", "")
    m = re.match(r"(fneus::\w+)", n)
    n = m.group(1) if m else n[:110]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{d:8.1f} us  grid {int(r['Grid_Size_X']):>9d}  {n}")
