#!/bin/bash
# round 5, call l: kernel statistics of the stage-1 step (small kernels at the step's ends)
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_l
rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o k --output-format csv -- python3 "$root/bench.py" --steps 16 --warmup 3 --no-cpu-baseline --no-fast-extra > /tmp/l.json 2>/dev/null
ks=$(find /tmp/prof_l -name '*kernel_stats.csv' | head -1)
python3 "$root/tools/summarize_prof.py" "$ks" 40 > "$out/r05_l_kernel_stats.txt"
grep "pack\|adam\|wn_back\|rowscale\|total GPU" "$out/r05_l_kernel_stats.txt" | cut -c1-110
cd $root; timeout 900 python3 -m pytest tests/test_hip_sdf.py tests/test_hip_properties.py tests/test_hip_refcolor.py -q -m gpu 2>&1 | tail -2
