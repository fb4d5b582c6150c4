#!/bin/bash
# round 6: colour backward on the r8 engine, fneus_color_out_dw with a reduce-scatter epilogue
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_hip_properties.py -q -m gpu -x -k "colour or color" 2>&1 | tail -8 | tee $out/r06_b_tests.txt
timeout 900 python3 -m pytest tests/test_hip_render.py tests/test_hip_backward.py tests/test_hip_training.py tests/test_hip_graph.py tests/test_hip_determinism.py -q -m gpu -x 2>&1 | tail -8 | tee -a $out/r06_b_tests.txt
python3 bench.py --no-cpu-baseline --no-fast-extra > $out/r06_b_bench.json 2> $out/r06_b_bench.err
python3 -c "
import json; d = json.loads(open('$out/r06_b_bench.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['value']); print(d['kernels_ms_per_step'])"
FNEUS_COL_BWD_R8=0 python3 bench.py --no-cpu-baseline --no-fast-extra > $out/r06_b_bench_tph.json 2>> $out/r06_b_bench.err
python3 -c "
import json; d = json.loads(open('$out/r06_b_bench_tph.json').read().strip().split('\n')[-1]); print('tph colour bwd', d['ms_per_step'], d['kernels_ms_per_step']['fneus_color_bwd'])"
