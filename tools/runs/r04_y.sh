#!/bin/bash
# NeRF products merged into the SDF launch (womask): tests + womask bench both ways
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests/test_hip_render.py tests/test_hip_nerf.py tests/test_hip_gemm_pp.py tests/test_hip_training.py tests/test_hip_graph.py tests/test_hip_determinism.py -q -m gpu -x 2>&1 | tail -5 | tee $out/r04_y_tests.txt
rm -f $out/r04_y_bench.txt
for m in 1 0 1 0; do
FNEUS_GEMM_MERGE=$m python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge $m', d['ms_per_step'], 'womask', d['womask_step']['ms_per_step'], 'womask256', d['womask_256_rays_step']['ms_per_step'])" | tee -a $out/r04_y_bench.txt
done
