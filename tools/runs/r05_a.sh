#!/bin/bash
# round 5, call a: the new / changed tests, the anchor A/B of fn_sincos (advisor finding 3), a baseline bench line on this round's box
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_hip_determinism.py tests/test_hip_rays.py "tests/test_hip_render.py::test_two_renders_one_backward_keep_every_weight_gradient" "tests/test_hip_render.py::test_colour_products_inside_the_sdf_launch_change_nothing" tests/test_hip_graph.py tests/test_hip_training.py -q -m gpu -x 2>&1 | tail -8 | tee $out/r05_a_tests.txt
echo "--- no anchor variant" | tee $out/r05_a_anchor.txt
FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_noanchor.so timeout 600 python3 -m pytest tests/test_hip_determinism.py -q -m gpu 2>&1 | tail -5 | tee -a $out/r05_a_anchor.txt
FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_noanchor.so timeout 300 python3 tools/experiments/r04/col_repro_dbg.py 2>&1 | tail -3 | tee -a $out/r05_a_anchor.txt
timeout 600 python3 bench.py --steps 50 --warmup 10 2>/dev/null | tail -1 > $out/r05_a_bench.json
python3 -c "
import json; d=json.load(open('$out/r05_a_bench.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d.get('kernels_ms_per_step'), indent=0))" | tee $out/r05_a_bench.txt
