#!/usr/bin/env python3
"""Per-kernel HIP-event times of one training step (eager launches): python tools/step_profile.py [wmask|womask|stage2|stage3] [parity|fast]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch

from fneus import ops
from fneus.trainer import Stage1Trainer, synthetic_batches, WMASK_MODEL

which = sys.argv[1] if len(sys.argv) > 1 else "wmask"
prec = ops.PREC_FAST if len(sys.argv) > 2 and sys.argv[2] == "fast" else ops.PREC_PARITY
dev = torch.device("cuda:0")
batches = synthetic_batches(8, 512, dev)
if which == "stage2":
    from fneus.trainer2 import Stage2Trainer
    tr = Stage2Trainer(dev, prec=prec)
    step = lambda b: tr.train_step(b)
elif which == "stage3":
    from fneus.trainer3 import Stage3Trainer
    tr = Stage3Trainer(dev, prec=prec)
    step = lambda b: tr.train_step(b)
else:
    conf = copy.deepcopy(WMASK_MODEL)
    if which == "womask":
        conf["neus_renderer"]["n_outside"] = 32
    tr = Stage1Trainer(dev, model_conf=conf, prec=prec, use_graph=False)
    bg = torch.ones(1, 3, device=dev) if which == "womask" else None
    step = lambda b: tr.train_step(b, cos_anneal_ratio=0.5, background_rgb=bg)
for b in batches[:4]:
    step(b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for b in batches[4:]:
    step(b)
e1.record()
torch.cuda.synchronize()
print(f"{which}: {e0.elapsed_time(e1) / 4:.3f} ms per eager step")
ops.profile_begin()
for b in batches[4:7]:
    step(b)
prof = ops.profile_end()
tot = 0.0
for name, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:32s} {n / 3:6.1f} launches/step {ms / 3:8.4f} ms/step")
    tot += ms / 3
print(f"  fneus kernels total {tot:.3f} ms/step")
