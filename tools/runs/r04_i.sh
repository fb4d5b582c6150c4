#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
for v in "" r8_ws6 r8_ws16; do
  if [ -n "$v" ]; then export FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_$v.so; fi
  timeout 300 python3 tools/experiments/r04/k2_rev_r8_time.py 2>&1 | tail -1
  timeout 300 python3 tools/experiments/r04/k3_r8_time.py 2>&1 | tail -2
done | tee $out/r04_i_time.txt
unset FNEUS_LIB
timeout 900 python3 -m pytest tests/test_hip_properties.py -x -q -m gpu -k r8 2>&1 | tail -3 | tee $out/r04_i_tests.txt
