#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_hip_properties.py -x -q -m gpu 2>&1 | tail -4 | tee $out/r04_h_tests.txt
timeout 1500 python3 -m pytest tests/test_hip_sdf.py tests/test_hip_backward.py tests/test_hip_render.py tests/test_hip_determinism.py tests/test_hip_graph.py tests/test_hip_training.py -x -q -m gpu 2>&1 | tail -4 | tee -a $out/r04_h_tests.txt
for v in "1 1" "0 0"; do set -- $v
FNEUS_K2_REV8=$1 FNEUS_K3_R8=$2 python3 bench.py --no-cpu-baseline --no-fast-extra --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('REV8=$1 K3_R8=$2', d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))" | tee -a $out/r04_h_bench.txt
done
