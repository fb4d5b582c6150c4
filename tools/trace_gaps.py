#!/usr/bin/env python3
"""Idle gaps of the GPU inside a replayed step, from a rocprofv3 --kernel-trace CSV:
    tools/trace_gaps.py <kernel_trace.csv> [steps]
Steps are delimited by the Adam kernel; for the last `steps` steps: busy time (union of the kernel intervals), idle time
and the largest gaps with the kernels either side."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
ends = ends[-(n_steps + 1):]
tot_busy = tot_idle = 0.0
gaps = defaultdict(lambda: [0, 0.0])
for a, b in zip(ends, ends[1:]):
    seg = rows[a + 1: b + 1]
    t_end = rows[a][1]
    for s, e, name in seg:
        if s > t_end:
            prev = [x for x in seg if x[1] == t_end]
            key = ((prev[0][2] if prev else rows[a][2])[:50], name[:50])
            gaps[key][0] += 1
            gaps[key][1] += (s - t_end) / 1e3
            tot_idle += (s - t_end) / 1e3
            tot_busy += (e - s) / 1e3
        else:
            tot_busy += max(0, e - max(s, t_end)) / 1e3
        t_end = max(t_end, e)
n = len(ends) - 1
print(f"{n} steps: step {(tot_busy + tot_idle) / n:.1f} us = busy {tot_busy / n:.1f} + idle {tot_idle / n:.1f}")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t / n:7.2f} us/step ({c / n:.1f} x)  {k[0]}  ->  {k[1]}")
