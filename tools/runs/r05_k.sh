#!/bin/bash
# round 5, call k: what gradient precision 2 costs, kernel by kernel
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for g in 1 2; do
  rm -rf /tmp/prof_g$g
  FNEUS_GPREC=$g rocprofv3 --kernel-trace --stats -d /tmp/prof_g$g -o k --output-format csv -- python3 "$root/bench.py" --steps 16 --warmup 3 --no-cpu-baseline --no-fast-extra > /tmp/g$g.json 2>/dev/null
  ks=$(find /tmp/prof_g$g -name '*kernel_stats.csv' | head -1)
  python3 "$root/tools/summarize_prof.py" "$ks" 14 > "$out/r05_k_gprec${g}_kernel_stats.txt"
  head -18 "$out/r05_k_gprec${g}_kernel_stats.txt" | cut -c1-120
done
