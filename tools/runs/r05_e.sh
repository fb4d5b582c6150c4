#!/bin/bash
# round 5, call e: the stage-2 / 3 MLPs on the fneus_mlp_* kernels: parity tests, stage tests, step times with and without
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_hip_mlp_rows.py -q -m gpu -s 2>&1 | grep "mlp_rows\|passed\|failed\|Error\|assert" | tee $out/r05_e_tests.txt
timeout 1200 python3 -m pytest tests/test_hip_stage2.py tests/test_hip_stage3.py -q -m gpu -x 2>&1 | tail -5 | tee -a $out/r05_e_tests.txt
for st in stage2 stage3; do
  for v in 1 0; do
    echo -n "FNEUS_MLP_ROWS=$v "; FNEUS_MLP_ROWS=$v python3 tools/stage_profile_run.py $st 40 2>&1 | tail -1
  done
done | tee $out/r05_e_times.txt
bash tools/collect_stage_profiles.sh r05_e stage2 stage3 > /dev/null 2>&1
head -5 $out/r05_e_stage2_kernel_stats.txt $out/r05_e_stage3_kernel_stats.txt
grep -c Cijk $out/r05_e_stage2_kernel_stats.txt $out/r05_e_stage3_kernel_stats.txt
