#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests/test_hip_stage3.py tests/test_hip_stage2.py tests/test_hip_runner.py -q -m gpu -x 2>&1 | tail -5 | tee $out/r04_z_tests.txt
python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'stage2', d['stage2_step']['ms_per_step'], 'stage3', d['stage3_step']['ms_per_step'])" | tee $out/r04_z_bench.txt
bash tools/collect_stage_profiles.sh r04_z stage3 2>&1 | grep -E "launches/step|per replayed"
