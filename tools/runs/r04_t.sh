#!/bin/bash
# observed values behind the bounds of the Adam-step test and the scene test (deterministic HIP runs), twice
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
for i in 1 2; do
python3 -m pytest tests/test_hip_render.py -q -m gpu -k adam_steps -s 2>&1 | grep -E "Adam step|passed|failed" | tee $out/r04_t_adam_$i.txt
python3 -m pytest tests/test_hip_scene.py -q -m gpu -k reconstruction -s 2>&1 | grep -E "first 10|first 3 windows|mean over|Chamfer|passed|failed|Error|assert" | tee $out/r04_t_scene_$i.txt
done
