#!/bin/bash
# round-4 baseline of the unchanged round-3 build on this round's box
set -u
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
python3 $root/bench.py --no-cpu-baseline > $out/r04_a_bench.json 2> $out/r04_a_bench.err; tail -c 600 $out/r04_a_bench.json
python3 $root/tools/torch_ops_in_step.py > $out/r04_a_torch_ops.txt 2>&1; tail -40 $out/r04_a_torch_ops.txt
WGS=256 python3 $root/tools/dbg_gemm_pp_time.py > $out/r04_a_gemm.txt 2>&1; cat $out/r04_a_gemm.txt
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace -d /tmp/prof_t -o t --output-format csv -- python3 $root/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-fast-extra --no-profile > /dev/null 2> $out/r04_a_rocprof_t.err
kt=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
python3 $root/tools/step_timeline.py "$kt" > $out/r04_a_step_timeline.txt 2>&1; cat $out/r04_a_step_timeline.txt
