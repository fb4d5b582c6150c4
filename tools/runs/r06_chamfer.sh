#!/bin/bash
# round 6: the Chamfer-L1 study at equal steps on this round's kernels at the DEFAULT gradient precision (2): 32 seeds x 2000 steps,
# HIP (deterministic) vs oracle
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 2700 python3 tests/checkers/chamfer_study.py --seeds 32 --steps 2000 --seed0 100 --gprec 2 --out $out/r06_chamfer.json 2>&1 | tail -3 | tee $out/r06_chamfer.txt
