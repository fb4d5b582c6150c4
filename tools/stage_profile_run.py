#!/usr/bin/env python3
"""N replayed steps of the stage-2 / stage-3 trainer for rocprofv3: python tools/stage_profile_run.py stage3 [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch

from fneus.trainer import synthetic_batches

which = sys.argv[1] if len(sys.argv) > 1 else "stage3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
step_kw = {}
if which == "stage2":
    from fneus.trainer2 import Stage2Trainer as T
    tr = T(dev, use_graph=True)
elif which == "womask":          # stage 1 with the background NeRF++ (womask.conf: + 32 outside samples per ray)
    import copy
    from fneus.trainer import Stage1Trainer, WMASK_MODEL
    conf = copy.deepcopy(WMASK_MODEL)
    conf["neus_renderer"]["n_outside"] = 32
    tr = Stage1Trainer(dev, model_conf=conf, use_graph=True)
    step_kw = dict(cos_anneal_ratio=0.5, background_rgb=torch.ones(1, 3, device=dev))
else:
    from fneus.trainer3 import Stage3Trainer as T
    tr = T(dev, use_graph=True)
batches = synthetic_batches(4, 512, dev)
for i in range(4):
    tr.train_step(batches[i], **step_kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps):
    tr.train_step(batches[i % 4], **step_kw)
e1.record()
torch.cuda.synchronize()
print(f"{which}: {e0.elapsed_time(e1) / steps:.3f} ms per replayed step")
