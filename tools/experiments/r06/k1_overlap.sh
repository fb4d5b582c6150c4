#!/bin/bash
# Round 6: does the vector work of the two-pass K1 hide behind its MFMAs when NOTHING else is in the stream?
# (r05/k1_parts.sh: MFMAs alone 76 us, vector work alone 78-83, full kernel 183 at 65 536 points: they add up.  Missing there:
# MFMAs + vector work WITHOUT the operand traffic.)  Variants of sdf_p2_kernels.hip, parts compiled out (times only):
#   build:  bash tools/experiments/r06/k1_overlap.sh build     (CPU container)
#   run:    bash tools/experiments/r06/k1_overlap.sh           (GPU box)
cd "$(dirname "$0")/../../.."
NOMEM="-DFNEUS_P2_NO_WEIGHTS -DFNEUS_P2_NO_LDSB -DFNEUS_P2_NO_LDSW"
declare -A V=(
  [k1_full]=""
  [k1_nomem]="$NOMEM"
  [k1_nomem_mfma]="$NOMEM -DFNEUS_P2_NO_VALU"
  [k1_nomem_valu]="$NOMEM -DFNEUS_P2_NO_MFMA"
  [k1_now]="-DFNEUS_P2_NO_WEIGHTS"
  [k1_noldsb]="-DFNEUS_P2_NO_LDSB"
  [k1_noldsw]="-DFNEUS_P2_NO_LDSW"
  [k1_nomem_cheap]="$NOMEM -DFNEUS_DBG_CHEAP_ACT"
)
if [ "$1" == "build" ]; then
  for v in "${!V[@]}"; do
    ( bash tools/experiments/build_variant.sh $v "${V[$v]}" sdf_p2_kernels.hip > /dev/null 2>&1 && echo built $v ) &
  done
  wait
  exit 0
fi
for v in k1_full k1_nomem k1_nomem_mfma k1_nomem_valu k1_nomem_cheap k1_now k1_noldsb k1_noldsw; do
  printf "%-16s " $v
  FNEUS_LIB=$PWD/factored-neus_amd/fneus/variants/libfneus_$v.so python3 tools/experiments/r06/k1_time.py 2>&1 | tail -1
done
