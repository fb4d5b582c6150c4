#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 600 python3 -m pytest tests/test_hip_properties.py -x -q -m gpu 2>&1 | tail -5 | tee $out/r04_d_tests.txt
for r8 in 0 1; do
FNEUS_K2_REV8=$r8 python3 bench.py --no-cpu-baseline --no-fast-extra --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('REV8=$r8', d['ms_per_step'], json.dumps(d['kernels_ms_per_step']), json.dumps(d.get('parity')))" | tee -a $out/r04_d_bench.txt
done
