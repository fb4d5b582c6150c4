#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
WGS=256 python3 tools/dbg_gemm_pp_time.py 2>&1 | grep gemm_pp | tee $out/r04_c_gemm.txt
timeout 900 python3 -m pytest tests/test_hip_gemm_pp.py tests/test_hip_loss.py tests/test_hip_graph.py tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -5 | tee $out/r04_c_tests.txt
python3 bench.py --no-cpu-baseline --no-fast-extra --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))" | tee $out/r04_c_bench.txt
