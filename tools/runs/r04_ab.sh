#!/bin/bash
# A/B of a library variant against the default build: tools/runs/r04_ab.sh NAME [reps]
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
name=$1; reps=${2:-2}
rm -f $out/r04_ab_$name.txt
for i in $(seq $reps); do
for lib in default $name; do
if [ $lib == default ]; then unset FNEUS_LIB; else export FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_$name.so; fi
python3 bench.py --no-cpu-baseline --no-fast-extra --steps 60 --warmup 5 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; print('$lib', round(d['ms_per_step'],4), {x:k[x] for x in ('fneus_sdf_fwd_grad','fneus_sdf_bwd','fneus_sdf_fwd','fneus_color_fwd','fneus_color_bwd') if x in k})" | tee -a $out/r04_ab_$name.txt
done
done
