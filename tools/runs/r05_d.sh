#!/bin/bash
# round 5, call d: render graph test, K1 variants at the sampler's size, bench
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 600 python3 -m pytest tests/test_hip_graph.py tests/test_hip_sdf.py -q -m gpu 2>&1 | tail -3 | tee $out/r05_d_tests.txt
python3 tools/experiments/r05/k1_h6_time.py 2>&1 | grep "seed 20" | tee $out/r05_d_k1.txt
for v in 31 3 32 4; do echo "FNEUS_K1_W8_BIG=$v"; FNEUS_K1_W8_BIG=$v python3 tools/experiments/r05/k1_h6_time.py 2>&1 | grep "seed 20 n 32768\|seed 20 n 65536" | sed 's/h6.*//'; done | tee -a $out/r05_d_k1.txt
timeout 900 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r05_d_bench.json
python3 -c "
import json; d=json.load(open('$out/r05_d_bench.json')); print(d['ms_per_step'], d['value'], 'fwd-only', d['forward_only_render']['ms_per_call'])" | tee $out/r05_d_bench.txt
