#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests/test_hip_loss.py tests/test_hip_render.py tests/test_hip_refcolor.py tests/test_hip_graph.py tests/test_hip_determinism.py tests/test_hip_training.py tests/test_hip_properties.py tests/test_hip_dp.py -x -q -m gpu 2>&1 | tail -4 | tee $out/r04_p_tests.txt
python3 bench.py --no-cpu-baseline --no-fast-extra --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))" | tee $out/r04_p_bench.txt
python3 tools/torch_ops_in_step.py 2>&1 | tail -9 | tee $out/r04_p_torch_ops.txt
timeout 300 python3 tools/experiments/r04/k2_rev_r8_time.py 2>&1 | tail -1
