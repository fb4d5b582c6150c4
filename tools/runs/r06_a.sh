#!/bin/bash
# round 6, first state: gradient precision 2 as the default (fneus_color_out_dw), frozen-network refresh in front of graph replays
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_hip_render.py tests/test_hip_backward.py tests/test_hip_training.py tests/test_hip_stage2.py tests/test_hip_graph.py tests/test_hip_determinism.py -q -m gpu -x 2>&1 | tail -8 | tee $out/r06_a_tests.txt
python3 bench.py --no-cpu-baseline --no-fast-extra > $out/r06_a_bench.json 2> $out/r06_a_bench.err
python3 -c "
import json; d = json.loads(open('$out/r06_a_bench.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['value']); print(d['kernels_ms_per_step'])"
FNEUS_GPREC=1 python3 bench.py --no-cpu-baseline --no-fast-extra --no-profile > $out/r06_a_bench_g1.json 2>> $out/r06_a_bench.err
python3 -c "
import json; d = json.loads(open('$out/r06_a_bench_g1.json').read().strip().split('\n')[-1]); print('gprec1', d['ms_per_step'], d['value'])"
