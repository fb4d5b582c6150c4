#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc passes: a few stage-1 train steps on the HIP backend."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops
from fneus.trainer import Stage1Trainer, synthetic_batches
prec = ops.PREC_PARITY if (len(sys.argv) < 2 or sys.argv[1] == "parity") else ops.PREC_FAST
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
tr = Stage1Trainer(dev, prec=prec)
for b in synthetic_batches(steps, 512, dev):
    tr.train_step(b)
torch.cuda.synchronize()
print("done")
