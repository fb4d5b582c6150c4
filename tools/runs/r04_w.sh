#!/bin/bash
# RefColor products merged too: tests + bench
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests/test_hip_render.py tests/test_hip_refcolor.py tests/test_hip_gemm_pp.py tests/test_hip_determinism.py tests/test_hip_training.py tests/test_hip_loss.py tests/test_hip_graph.py -q -m gpu -x 2>&1 | tail -8 | tee $out/r04_w_tests.txt
rm -f $out/r04_w_bench.txt
for m in 1 0 1 0; do
FNEUS_GEMM_MERGE=$m python3 bench.py --no-cpu-baseline --no-fast-extra --steps 60 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge $m', d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if 'gemm' in k})" | tee -a $out/r04_w_bench.txt
done
