"""Which torch matmuls run in one eager stage-3 step (shapes and GPU time): python tools/experiments/r03/stage3_gemm_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from fneus.trainer import synthetic_batches
from fneus.trainer3 import Stage3Trainer
dev = torch.device("cuda:0")
tr = Stage3Trainer(dev, use_graph=False)
b = synthetic_batches(2, 512, dev)
tr.train_step(b[0]); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr._fixed_shape_step(b[1]); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if any(k in e.key for k in ("mm", "matmul", "linear", "bmm", "einsum"))]
for e in sorted(rows, key=lambda e: -e.device_time_total)[:14]:
    print(f"{e.key:28s} n={e.count:3d} gpu {e.device_time_total:8.1f} us  {str(e.input_shapes)[:110]}")
