#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_hip_bench.py tests/test_hip_properties.py tests/test_hip_sdf.py tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -4 | tee $out/r04_o_tests.txt
bash tools/collect_profiles.sh r04_a 2>&1 | tail -60
bash tools/collect_stage_profiles.sh r04_a 2>&1 | tail -30
