#!/bin/bash
# round 6: the cotangent chains on bf16 activations (FNEUS_BWD_XHI=1, the new default) against hi + lo activations (=0): K3 time,
# the double-backward test's errors (random cotangents), the golden-gradient errors of the seven fixtures, the step
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
for v in 1 0; do
  export FNEUS_BWD_XHI=$v
  echo "== FNEUS_BWD_XHI=$v"
  python3 tools/dbg_k23_time.py
  python3 -m pytest tests/test_hip_backward.py -q -m gpu -s -k "double_backward" 2>&1 | grep -E "gprec=1 dW|passed|failed" | awk '{print}' | sort | tail -22
  python3 -m pytest tests/test_hip_render.py -q -m gpu -s -k "test_loss_and_gradients and fused_loss and grad_mixed" 2>&1 | grep -E "worst relative|passed|failed|Error|assert"
done 2>&1 | tee $out/r06_xhi.txt
for v in 1 0; do
  export FNEUS_BWD_XHI=$v
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fast-extra --no-profile 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('XHI=$v', d['ms_per_step'], d.get('kernels_ms_per_step'))"
done 2>&1 | tee -a $out/r06_xhi.txt
