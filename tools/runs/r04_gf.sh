#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
rm -f $out/r04_gf.txt
for rep in 1 2; do
for fl in 16 12 8 20; do
FNEUS_GEMM_FLOOR=$fl python3 bench.py --no-cpu-baseline --no-fast-extra --steps 60 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('floor $fl', round(d['ms_per_step'],4), d['kernels_ms_per_step']['fneus_dw_gemm_pp:sdf+color'])" | tee -a $out/r04_gf.txt
done
done
