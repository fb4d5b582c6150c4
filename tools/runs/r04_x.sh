#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests/test_hip_loss.py tests/test_hip_render.py tests/test_hip_training.py tests/test_hip_determinism.py tests/test_hip_dp.py -q -m gpu -x 2>&1 | tail -5 | tee $out/r04_x_tests.txt
python3 bench.py --no-cpu-baseline --no-fast-extra --steps 60 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))" | tee $out/r04_x_bench.txt
