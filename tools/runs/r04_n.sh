#!/bin/bash
# A/B of the r8 kernels: current build against the first NH build (commit 96165ac), interleaved, 3 rounds
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
for rep in 1 2 3; do
for v in "" r8_prev; do
  if [ -n "$v" ]; then export FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_$v.so; else unset FNEUS_LIB; fi
  echo "== ${v:-current}"
  timeout 300 python3 tools/experiments/r04/k2_rev_r8_time.py 2>&1 | tail -1
  timeout 300 python3 tools/experiments/r04/k3_r8_time.py 2>&1 | tail -1
done; done | tee $out/r04_n_ab.txt
unset FNEUS_LIB
bash tools/runs/r04_m.sh
