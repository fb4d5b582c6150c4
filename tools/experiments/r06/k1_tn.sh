#!/bin/bash
# Round 6, review item 5: K1's two-pass kernel with a wave owning TWO output tiles against one B read (4 waves x 512 registers, half the LDS
# reads per MFMA: FNEUS_K1_W8_BIG=3) beside the shipped form (8 waves, one tile each: 31) -- time, matrix-pipe busy, issue stalls, LDS counters.
root=$(cd "$(dirname "$0")/../../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for big in 31 3; do
  export FNEUS_K1_W8_BIG=$big
  python3 $root/tools/experiments/r06/k1_one.py 8 2>&1 | tail -1
  rm -rf /tmp/k1tn_a /tmp/k1tn_b
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES \
      -d /tmp/k1tn_a -o a --output-format csv -- python3 $root/tools/experiments/r06/k1_one.py 2 > /dev/null 2> $out/k1tn_a_$big.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
      -d /tmp/k1tn_b -o b --output-format csv -- python3 $root/tools/experiments/r06/k1_one.py 2 > /dev/null 2> $out/k1tn_b_$big.err
  python3 $root/tools/pmc_summary.py /tmp/k1tn_a /tmp/k1tn_b | grep -A18 "^sdf_fwd_p2_kernel" | head -24
done
