#!/bin/bash
root=/root/repo; out=$root/gpurun_out
export TMPDIR=/tmp; cd /tmp
for v in default ${VARIANT:-p2_nostore}; do
  if [ $v != default ]; then export FNEUS_LIB=$root/factored-neus_amd/fneus/variants/libfneus_$v.so; fi
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o k --output-format csv -- python3 "$root/bench.py" --steps 16 --warmup 3 --no-cpu-baseline --no-fast-extra --no-profile > $out/ab_lib_$v.json 2> /dev/null
  ks=$(find /tmp/prof_ab -name '*kernel_stats.csv' | head -1)
  echo "== $v"
  python3 "$root/tools/summarize_prof.py" "$ks" 14 | sed -n 4,13p | cut -c1-100
done
