#!/bin/bash
# round 6: kernel statistics of the replayed step under rocprofv3 for two settings of an environment switch: r06_ab.sh VAR A B
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
var=$1; shift
export TMPDIR=/tmp; cd /tmp
for v in "$@"; do
  export $var=$v
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o k --output-format csv -- python3 "$root/bench.py" --steps 16 --warmup 3 --no-cpu-baseline --no-fast-extra --no-profile > $out/ab_${var}_$v.json 2> /dev/null
  ks=$(find /tmp/prof_ab -name '*kernel_stats.csv' | head -1)
  echo "== $var=$v  $(python3 -c "import json; print(round(json.loads(open('$out/ab_${var}_$v.json').read().strip().split(chr(10))[-1])['ms_per_step'], 4))") ms per step under the profiler"
  python3 "$root/tools/summarize_prof.py" "$ks" 14 | sed -n 4,16p
done
