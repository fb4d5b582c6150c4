#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
rm -f $out/r04_s_bench.txt
for hb in 0 1; do
FNEUS_K7_HB=$hb python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K7_HB $hb', d['ms_per_step'])
for k in ('womask_step','womask_256_rays_step'): print(k, d[k].get('ms_per_step', d[k].get('ms_per_call')))" | tee -a $out/r04_s_bench.txt
done
bash tools/collect_stage_profiles.sh r04_s womask 2>&1 | tail -26
