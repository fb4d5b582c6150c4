#!/bin/bash
# round 5, call g: grouped MLPs: tests, stage tests, stage step times, launch counts
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_hip_mlp_rows.py tests/test_hip_stage2.py tests/test_hip_stage3.py -q -m gpu -x 2>&1 | tail -5 | tee $out/r05_g_tests.txt
for st in stage2 stage3; do
  python3 tools/stage_profile_run.py $st 40 2>&1 | tail -1
done | tee $out/r05_g_times.txt
bash tools/collect_stage_profiles.sh r05_g stage2 stage3 > /dev/null 2>&1
head -3 $out/r05_g_stage2_kernel_stats.txt $out/r05_g_stage3_kernel_stats.txt
