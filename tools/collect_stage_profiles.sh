#!/bin/bash
# rocprofv3 kernel statistics of the stage-2 / stage-3 / womask steps (replayed hipGraphs) into gpurun_out/<tag>_<stage>_kernel_stats.txt:
#   tools/collect_stage_profiles.sh r03_a [stage ...]
set -u
tag=${1:-r03_x}
shift || true
stages=${@:-stage2 stage3 womask}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for st in $stages; do
  rm -rf /tmp/prof_$st
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$st -o k --output-format csv -- python3 "$root/tools/stage_profile_run.py" $st 20 \
      > "$out/${tag}_${st}_run.txt" 2> "$out/${tag}_${st}_rocprof.err"
  ks=$(find /tmp/prof_$st -name '*kernel_stats.csv' | head -1)
  python3 "$root/tools/summarize_prof.py" "$ks" 24 > "$out/${tag}_${st}_kernel_stats.txt"
  cat "$out/${tag}_${st}_run.txt" | tail -1
  head -24 "$out/${tag}_${st}_kernel_stats.txt"
done
bad=0
for st in $stages; do
    f="$out/${tag}_${st}_kernel_stats.txt"
    if [ ! -s "$f" ] || grep -q "^Traceback" "$f"; then
        echo "collect_stage_profiles: BROKEN ARTEFACT $f" >&2
        bad=1
    fi
done
exit $bad
