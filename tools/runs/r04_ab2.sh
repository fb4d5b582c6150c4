#!/bin/bash
# A/B of a library variant on the steps that run K1-type kernels most: stage 2, stage 3, the headline step
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
name=$1; reps=${2:-2}
rm -f $out/r04_ab2_$name.txt
for i in $(seq $reps); do
for lib in default $name; do
if [ $lib == default ]; then unset FNEUS_LIB; else export FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_$name.so; fi
python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; print('$lib', round(d['ms_per_step'],4), 'K1', k['fneus_sdf_fwd'], 'stage2', round(d['stage2_step']['ms_per_step'],3), 'stage3', round(d['stage3_step']['ms_per_step'],3), 'fwd', round(d['forward_only_render']['ms_per_call'],3))" | tee -a $out/r04_ab2_$name.txt
done
done
