#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short text table (committed under profiles/)."""
import csv
import sys


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    ours = [r for r in rows if "fneus::" in r["Name"]]
    t_ours = sum(float(r["TotalDurationNs"]) for r in ours)
    print(f"# source: {path}")
    print(f"# total GPU kernel time {total / 1e6:.2f} ms over the run ({steps} steps incl. warm-up) -> {total / 1e6 / steps:.3f} ms/step")
    print(f"# fneus HIP kernels {t_ours / 1e6:.2f} ms ({100 * t_ours / total:.1f} %), PyTorch/other {100 - 100 * t_ours / total:.1f} % "
          f"in {sum(int(r['Calls']) for r in rows if 'fneus::' not in r['Name']) / steps:.0f} launches/step")
    print(f"{'kernel':60s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>9s} {'%':>6s}")
    for r in rows[:25]:
        name = r["Name"].replace("void ", "")
        name = name.split("(")[0][:60]
        print(f"{name:60s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.1f} {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
