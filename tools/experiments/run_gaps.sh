export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
cd /tmp
for mode in plain early noearly; do
  unset FNEUS_DP_SINGLE FNEUS_DP_EARLY
  [ $mode = early ] && export FNEUS_DP_SINGLE=1
  [ $mode = noearly ] && export FNEUS_DP_SINGLE=1 FNEUS_DP_EARLY=0
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace -d /tmp/prof_$mode -o t --output-format csv -- python3 $root/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-fast-extra --no-profile > /tmp/$mode.json 2> /tmp/$mode.err
  kt=$(find /tmp/prof_$mode -name '*kernel_trace.csv' | head -1)
  echo "== $mode"; grep -o '"ms_per_step": [0-9.]*' /tmp/$mode.json
  python3 $root/tools/trace_gaps.py $kt 20
done
