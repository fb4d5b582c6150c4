"""Round 6: the stage-3 step (one hipGraph replay per step) timed as bench.py times it, three times over"""
import os, sys, time
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "factored-neus_amd"))
import torch
from fneus import ops
from fneus.trainer import synthetic_batches
from fneus.trainer3 import Stage3Trainer
dev = torch.device("cuda:0")
sb = synthetic_batches(14, 512, dev, rank=0)
for rep in range(3):
    tr = Stage3Trainer(dev, prec=ops.PREC_PARITY, use_graph=True)
    for b in sb[:4]: tr.train_step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in sb[4:]: tr.train_step(b)
    torch.cuda.synchronize(); print("stage-3 step %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    del tr
