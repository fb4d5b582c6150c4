#!/bin/bash
# Collect one round's judged artefacts on the GPU box into gpurun_out/<tag>_* (copy the ones to keep into profiles/):
#   tools/collect_profiles.sh r01_d
# bench line, the same command under rocprofv3 --kernel-trace --stats, one eager step's timeline, and the HBM traffic
# counters (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, no trace domains alongside).
set -u
tag=${1:-r01_x}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
python3 "$root/bench.py" > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
tail -1 "$out/${tag}_bench.json"
rm -rf /tmp/prof_k /tmp/prof_t /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k --output-format csv -- python3 "$root/bench.py" --steps 16 --warmup 3 --no-cpu-baseline --no-fast-extra \
    > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_rocprof.err"
ks=$(find /tmp/prof_k -name '*kernel_stats.csv' | head -1)
python3 "$root/tools/summarize_prof.py" "$ks" 22 > "$out/${tag}_kernel_stats.txt"
head -16 "$out/${tag}_kernel_stats.txt"
rocprofv3 --kernel-trace -d /tmp/prof_t -o t --output-format csv -- python3 "$root/bench.py" --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-fast-extra --no-profile \
    > /dev/null 2> "$out/${tag}_rocprof_t.err"
kt=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
python3 "$root/tools/step_timeline.py" "$kt" > "$out/${tag}_step_timeline.txt" 2>&1 || echo "step_timeline FAILED" >&2
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f --output-format csv -- python3 "$root/tools/pmc_run.py" parity 2 > /dev/null 2> "$out/${tag}_pmc_f.err"
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w --output-format csv -- python3 "$root/tools/pmc_run.py" parity 2 > /dev/null 2> "$out/${tag}_pmc_w.err"
python3 "$root/tools/pmc_summary.py" --steps 2 --json "$out/${tag}_traffic.json" /tmp/pmc_f /tmp/pmc_w > "$out/${tag}_pmc_fetch_write.txt"
grep -A2 -E "^(sdf_bwd|sdf_fwd_grad|dw_gemm)|whole step" "$out/${tag}_pmc_fetch_write.txt"
# matrix-pipe utilisation per kernel: SQ counters in their own pass (8 SQ slots)
rm -rf /tmp/pmc_s
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES -d /tmp/pmc_s -o s --output-format csv -- python3 "$root/tools/pmc_run.py" parity 2 > /dev/null 2> "$out/${tag}_pmc_s.err"
python3 "$root/tools/pmc_summary.py" /tmp/pmc_s > "$out/${tag}_sq.txt"
head -40 "$out/${tag}_sq.txt"

# gate: an artefact that holds a Python traceback (or an empty summary) is not evidence -- fail the collection
bad=0
for f in "$out/${tag}_bench.json" "$out/${tag}_bench_under_rocprof.json" "$out/${tag}_kernel_stats.txt" "$out/${tag}_step_timeline.txt" \
         "$out/${tag}_pmc_fetch_write.txt" "$out/${tag}_traffic.json" "$out/${tag}_sq.txt"; do
    if [ ! -s "$f" ] || grep -q -E "^Traceback|^step_timeline:" "$f"; then
        echo "collect_profiles: BROKEN ARTEFACT $f" >&2
        bad=1
    fi
done
exit $bad
