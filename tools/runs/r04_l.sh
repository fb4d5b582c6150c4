#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 300 python3 tools/experiments/r04/k3_r8_time.py 65536 3 2>&1 | tail -4 | tee $out/r04_l_k3_gp3.txt
for g in 1 3; do
FNEUS_GPREC=$g python3 bench.py --no-cpu-baseline --no-fast-extra --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPREC=$g', d['ms_per_step'], json.dumps(d['kernels_ms_per_step']), json.dumps(d.get('floor',{}).get('step_floor_ms')))" | tee -a $out/r04_l_bench.txt
done
python3 tools/torch_ops_in_step.py 2>&1 | tail -14 | tee $out/r04_l_torch_ops.txt
timeout 2200 python3 -m pytest tests -q -m gpu --durations=25 2>&1 | tail -45 | tee $out/r04_l_suite_durations.txt
