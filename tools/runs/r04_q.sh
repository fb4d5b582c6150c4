#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))
for k in ('stage2_step','stage3_step','womask_step','womask_256_rays_step','forward_only_render'): print(k, d[k].get('ms_per_step', d[k].get('ms_per_call')))
print(json.dumps(d.get('parity')))" | tee $out/r04_q_bench.txt
timeout 2200 python3 -m pytest tests -q -m gpu -x --deselect tests/test_hip_scene.py::test_chamfer_at_equal_steps_hip_vs_oracle_over_seeds 2>&1 | tail -6 | tee $out/r04_q_tests.txt
