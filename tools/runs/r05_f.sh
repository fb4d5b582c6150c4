#!/bin/bash
# round 5, call f: per-launch times of the fneus_mlp_* kernels, the tests again, the stage steps
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
python3 tools/experiments/r05/mlp_rows_time.py 2>&1 | tee $out/r05_f_rows_time.txt
timeout 900 python3 -m pytest tests/test_hip_mlp_rows.py -q -m gpu 2>&1 | tail -2 | tee $out/r05_f_tests.txt
for st in stage2 stage3; do
  python3 tools/stage_profile_run.py $st 40 2>&1 | tail -1
done | tee $out/r05_f_times.txt
