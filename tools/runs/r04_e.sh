#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 300 python3 tools/experiments/r04/k2_rev_r8_dbg.py 2>&1 | tail -30 | tee $out/r04_e_dbg.txt
