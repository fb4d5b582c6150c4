#!/bin/bash
# GEMM part timings: bias column sums / atomics epilogue compiled out
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd /tmp
for v in "" gpp_nobias gpp_noatomic gpp_nobias_noatomic; do
  if [ -n "$v" ]; then export FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_$v.so; fi
  echo "== ${v:-default}"; WGS=256 python3 $root/tools/dbg_gemm_pp_time.py 2>&1 | grep gemm_pp
done > $out/r04_b_gemm_parts.txt
cat $out/r04_b_gemm_parts.txt
