#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_r8_stamps.so timeout 300 python3 tools/experiments/r04/r8_stamps.py 2>&1 | grep "_r8 NH" | tail -6 | tee $out/r04_m_stamps.txt
timeout 300 python3 tools/experiments/r03/torch_ops_in_stage3_step.py 2>&1 | tail -140 > $out/r04_m_stage3_ops.txt
timeout 300 python3 tools/experiments/r03/torch_ops_in_stage2_step.py 2>&1 | tail -90 > $out/r04_m_stage2_ops.txt
tail -3 $out/r04_m_stage3_ops.txt $out/r04_m_stage2_ops.txt
