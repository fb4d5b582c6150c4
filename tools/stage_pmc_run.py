#!/usr/bin/env python3
"""A few EAGER steps of the stage-2 / stage-3 trainer for rocprofv3 --pmc passes: python tools/stage_pmc_run.py stage3 [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch

from fneus.trainer import synthetic_batches

which = sys.argv[1] if len(sys.argv) > 1 else "stage3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
if which == "stage2":
    from fneus.trainer2 import Stage2Trainer as T
else:
    from fneus.trainer3 import Stage3Trainer as T
tr = T(dev, use_graph=False)
for b in synthetic_batches(steps, 512, dev):
    tr._fixed_shape_step(b)          # the fixed-shape step (what the replayed graph runs), launched eagerly
torch.cuda.synchronize()
print("done")
