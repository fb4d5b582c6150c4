#!/bin/bash
# background selection (fneus_outside_select): its tests, the womask / NeRF / GEMM tests, then the bench both ways
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1200 python3 -m pytest tests/test_hip_render.py tests/test_hip_nerf.py tests/test_hip_gemm_pp.py tests/test_hip_determinism.py -q -m gpu -x 2>&1 | tail -8 | tee $out/r04_r_tests.txt
for sel in 1 0; do
FNEUS_BG_SELECT=$sel python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('select $sel', d['ms_per_step'])
for k in ('womask_step','womask_256_rays_step'): print(k, d[k].get('ms_per_step', d[k].get('ms_per_call')), json.dumps(d[k].get('kernels_ms_per_step', {})))" | tee -a $out/r04_r_bench.txt
done
