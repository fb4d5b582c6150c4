#!/bin/bash
# PMC counters of the stage-2 / stage-3 steps (eager fixed-shape steps): FETCH_SIZE, WRITE_SIZE and the SQ set in separate passes
#   tools/collect_stage_pmc.sh r06_x [stage ...]   ->  gpurun_out/<tag>_<stage>_pmc.json / .txt
set -u
tag=${1:-r06_x}
shift || true
stages=${@:-stage2 stage3}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for st in $stages; do
  rm -rf /tmp/spmc_f /tmp/spmc_w /tmp/spmc_s
  rocprofv3 --pmc FETCH_SIZE -d /tmp/spmc_f -o f --output-format csv -- python3 "$root/tools/stage_pmc_run.py" $st 2 > /dev/null 2> "$out/${tag}_${st}_pmc_f.err"
  rocprofv3 --pmc WRITE_SIZE -d /tmp/spmc_w -o w --output-format csv -- python3 "$root/tools/stage_pmc_run.py" $st 2 > /dev/null 2> "$out/${tag}_${st}_pmc_w.err"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES \
      -d /tmp/spmc_s -o s --output-format csv -- python3 "$root/tools/stage_pmc_run.py" $st 2 > /dev/null 2> "$out/${tag}_${st}_pmc_s.err"
  python3 "$root/tools/pmc_kernels_json.py" --steps 2 --out "$out/${tag}_${st}_pmc.json" /tmp/spmc_f /tmp/spmc_w /tmp/spmc_s | tee "$out/${tag}_${st}_pmc.txt"
done
