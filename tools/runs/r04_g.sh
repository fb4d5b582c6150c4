#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
for nh in 2 4; do
  export FNEUS_R8_NH=$nh
  FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_r8_stamps.so timeout 300 python3 tools/experiments/r04/r8_stamps.py 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 300 python3 tools/experiments/r04/k2_rev_r8_time.py 2>&1 | tail -1
  timeout 300 python3 tools/experiments/r04/k3_r8_time.py 2>&1 | tail -2
done | tee $out/r04_g_time.txt
unset FNEUS_R8_NH
timeout 900 python3 -m pytest tests/test_hip_properties.py -x -q -m gpu -k "r8" 2>&1 | tail -8 | tee $out/r04_g_tests.txt
