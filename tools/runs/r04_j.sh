#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
for w in 4 8 4 8; do echo "waves $w"; FNEUS_GEMM_WAVES=$w WGS=256 python3 tools/dbg_gemm_pp_time.py 2>&1 | grep gemm_pp; done | tee $out/r04_j_gemm.txt
for w in 4 8; do FNEUS_GEMM_WAVES=$w timeout 600 python3 -m pytest tests/test_hip_gemm_pp.py tests/test_hip_determinism.py -x -q -m gpu 2>&1 | tail -2; done | tee $out/r04_j_tests.txt
for w in 4 8; do
FNEUS_GEMM_WAVES=$w python3 bench.py --no-cpu-baseline --no-fast-extra --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GEMM_WAVES=$w', d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))" | tee -a $out/r04_j_bench.txt
done
