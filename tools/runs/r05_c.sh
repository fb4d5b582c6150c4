#!/bin/bash
# round 5, call c: h6 prototype tests + the bench line with its extra
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 600 python3 -m pytest tests/test_hip_sdf.py tests/test_hip_stage2.py -q -m gpu -x 2>&1 | tail -4 | tee $out/r05_c_tests.txt
FNEUS_K1_H6=1 timeout 600 python3 -m pytest tests/test_hip_stage2.py tests/test_hip_training.py "tests/test_hip_render.py::test_render_end_to_end" -q -m gpu 2>&1 | tail -6 | tee $out/r05_c_tests_h6.txt
timeout 900 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r05_c_bench.json
python3 -c "
import json; d=json.load(open('$out/r05_c_bench.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d.get('h6_products_prototype'), indent=1)); print('stage2', d['stage2_step']['ms_per_step'], 'stage3', d['stage3_step']['ms_per_step'], 'womask', d['womask_step']['ms_per_step'])" | tee $out/r05_c_bench.txt
