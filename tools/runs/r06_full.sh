#!/bin/bash
# round 6: the whole GPU suite, then the judged artefacts of the state it ran on (tag = $1): bench line, rocprofv3 kernel stats,
# step timeline, FETCH / WRITE / SQ counters of the stage-1 step, kernel stats and counters of the stage-2 / stage-3 steps
tag=${1:-r06_x}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1700 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $out/${tag}_gpu_tests.txt
bash tools/collect_profiles.sh $tag > $out/${tag}_collect.log 2>&1; echo "collect rc $?" | tee -a $out/${tag}_gpu_tests.txt
bash tools/collect_stage_profiles.sh $tag > $out/${tag}_collect_stage.log 2>&1; echo "collect stage rc $?" | tee -a $out/${tag}_gpu_tests.txt
bash tools/collect_stage_pmc.sh $tag > $out/${tag}_collect_stage_pmc.log 2>&1; echo "collect stage pmc rc $?" | tee -a $out/${tag}_gpu_tests.txt
tail -1 $out/${tag}_bench.json | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); print({k: d[k].get('ms_per_step') for k in ('womask_step','stage2_step','stage3_step','exact_gradients_gprec3','bf16_gradient_planes_gprec1','fast_bf16','womask_256_rays_step') if k in d}); print(d.get('roofline')); print(d.get('box'))"
