#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 300 python3 tools/experiments/r04/k3_r8_time.py 2>&1 | grep -v amdgpu.ids | tee $out/r04_f_time.txt
timeout 600 python3 -m pytest tests/test_hip_properties.py -x -q -m gpu -k "k3_r8" 2>&1 | tail -15 | tee $out/r04_f_tests.txt
timeout 900 python3 -m pytest tests/test_hip_backward.py tests/test_hip_render.py tests/test_hip_determinism.py -x -q -m gpu 2>&1 | tail -8 | tee -a $out/r04_f_tests.txt
for r8 in 0 1; do
FNEUS_K3_R8=$r8 python3 bench.py --no-cpu-baseline --no-fast-extra --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K3_R8=$r8', d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))" | tee -a $out/r04_f_bench.txt
done
