#!/usr/bin/env python3
"""Step time of the stage-2 / stage-3 trainers: one graph (single GPU), data-parallel eager, data-parallel chain of graphs
(one rank: the collectives are identities unless FNEUS_DP_SINGLE=1 + a process group)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus.trainer import synthetic_batches
from fneus.trainer2 import Stage2Trainer
from fneus.trainer3 import Stage3Trainer
dev = torch.device("cuda:0")
batches = synthetic_batches(30, 512, dev)
for name, cls in (("stage 2", Stage2Trainer), ("stage 3", Stage3Trainer)):
    for label, kw in (("one graph", dict(use_graph=True)), ("data parallel, eager", dict(distributed=True)),
                      ("data parallel, graph chain", dict(distributed=True, use_graph=True))):
        tr = cls(dev, seed=3, **kw)
        for b in batches[:6]:
            tr.train_step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches[6:]:
            tr.train_step(b)
        torch.cuda.synchronize()
        print(f"{name}: {label:28s} {(time.perf_counter() - t0) / 24 * 1e3:7.2f} ms per step"
              + (f"  ({len(tr._seg.graphs)} graphs)" if tr._seg is not None else ""))
        del tr
