#!/bin/bash
# round 5, call b: the tightened scene / Chamfer / depth-agreement bounds
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd $root
timeout 1500 python3 -m pytest tests/test_hip_scene.py "tests/test_hip_render.py::test_render_end_to_end" -q -m gpu -s 2>&1 | grep -v "^\s*$" | tail -60 | tee $out/r05_b_tests.txt
