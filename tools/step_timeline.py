#!/usr/bin/env python3
"""Phase breakdown of ONE train step from a rocprofv3 --kernel-trace CSV: fneus kernels by name, the PyTorch kernels
between them as groups (count, busy time, idle gaps).  Usage: step_timeline.py kernel_trace.csv [top_n_torch]"""
import csv
import re
import sys
from collections import Counter


def short(n):
    n = n.replace("void ", "")
    m = re.match(r"(fneus::\w+)", n)
    if m:
        return m.group(1)
    return re.sub(r"at::native::", "", n)[:90]


def main(path, top=25):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at the first weight-norm fold of a burst: one `rowscale_multi_kernel` launch for all networks
    # (fneus_refresh_multi), or one `rowscale_kernel` per packed network with FNEUS_PACK_BATCH=0
    idx = [i for i, r in enumerate(rows) if re.search(r"rowscale(_multi)?_kernel", r["Kernel_Name"])]
    starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 8]
    if len(starts) < 2:
        sys.exit(f"step_timeline: found {len(starts)} step starts (rowscale launches) in {path}: need at least two steps")
    a, b = starts[-2], starts[-1]
    seg = rows[a:b]
    print(f"# one step: {len(seg)} kernels, span {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6:.3f} ms")
    out, cur, prev_end = [], [0, 0.0, 0.0], None
    torch_names = Counter()
    torch_time = Counter()
    for r in seg:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = 0 if prev_end is None else max(0, (int(r["Start_Timestamp"]) - prev_end) / 1e3)
        prev_end = int(r["End_Timestamp"])
        nm = short(r["Kernel_Name"])
        if nm.startswith("fneus::"):
            if cur[0]:
                out.append((f"  torch x{cur[0]}", cur[1], cur[2]))
                cur = [0, 0.0, 0.0]
            out.append((nm, d, gap))
        else:
            cur[0] += 1
            cur[1] += d
            cur[2] += gap
            torch_names[nm] += 1
            torch_time[nm] += d
    if cur[0]:
        out.append((f"  torch x{cur[0]}", cur[1], cur[2]))
    busy = sum(d for _, d, _ in out)
    for n, d, g in out:
        print(f"{n:40s} busy {d:8.1f} us   gaps {g:7.1f} us")
    print(f"# busy {busy / 1e3:.3f} ms, fneus {sum(d for n, d, _ in out if n.startswith('fneus')) / 1e3:.3f} ms, "
          f"torch {sum(torch_time.values()) / 1e3:.3f} ms in {sum(torch_names.values())} kernels")
    print("# PyTorch kernels by total time")
    for nm, t in torch_time.most_common(top):
        print(f"{torch_names[nm]:5d} x {t / torch_names[nm]:6.1f} us = {t:7.1f} us  {nm}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
