#!/bin/bash
# round 6: quick state check -- colour property tests, headline bench, kernel stats of the replayed step under rocprofv3
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
tag=${1:-r06_c}
cd $root
timeout 900 python3 -m pytest tests/test_hip_properties.py -q -m gpu -x -k "colour or color" 2>&1 | tail -4 | tee $out/${tag}_tests.txt
python3 bench.py --no-cpu-baseline --no-fast-extra > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python3 -c "
import json; d = json.loads(open('$out/${tag}_bench.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['value']); print(d['kernels_ms_per_step'])"
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k --output-format csv -- python3 "$root/bench.py" --steps 16 --warmup 3 --no-cpu-baseline --no-fast-extra > /dev/null 2> $out/${tag}_rocprof.err
ks=$(find /tmp/prof_k -name '*kernel_stats.csv' | head -1)
python3 "$root/tools/summarize_prof.py" "$ks" 22 > "$out/${tag}_kernel_stats.txt"
head -30 "$out/${tag}_kernel_stats.txt"
