#!/usr/bin/env python3
"""Train-step throughput against the ray batch per GPU (the reference's configurations use 512; config 5 uses 2048 global)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops
from fneus.trainer import Stage1Trainer, synthetic_batches
dev = torch.device("cuda:0")
for B in (256, 512, 1024, 2048, 4096):
    tr = Stage1Trainer(dev, prec=ops.PREC_PARITY, use_graph=True)
    bs = synthetic_batches(6, B, dev)
    for i in range(5):
        tr.train_step(bs[i % 6])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for i in range(n):
        tr.train_step(bs[i % 6])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B = {B:5d} rays x 128 samples: {dt * 1e3:7.3f} ms/step  {B * 128 / dt / 1e7:.3f}e7 ray-samples/s  "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    del tr
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
