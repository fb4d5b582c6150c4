#!/usr/bin/env python3
"""Which PyTorch (aten) ops still launch kernels inside one eager train step, in order, with their input shapes:
python tools/torch_ops_in_step.py [stage1 | stage2 | stage3]   (stages 2 / 3: the fixed-shape step the graphs replay)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from fneus.trainer import Stage1Trainer, synthetic_batches
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "stage1"
if which == "stage2":
    from fneus.trainer2 import Stage2Trainer
    tr = Stage2Trainer(dev, use_graph=False)
    step = tr._fixed_shape_step
elif which == "stage3":
    from fneus.trainer3 import Stage3Trainer
    tr = Stage3Trainer(dev, use_graph=False)
    step = tr._fixed_shape_step
else:
    tr = Stage1Trainer(dev, use_graph=False)
    step = tr.train_step
bs = synthetic_batches(4, 512, dev)
for b in bs[:3]:
    step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(bs[3])
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
# keep leaf aten ops that launched a kernel
out = []
for e in evs:
    if e.kernels:
        site = next((f for f in (e.stack or []) if "/factored-neus_amd/" in f or "/bench.py" in f), "")
        site = site.split("/factored-neus_amd/")[-1][:60]
        out.append((e.time_range.start, e.name, [k.name[:60] for k in e.kernels], e.input_shapes, site))
out.sort()
seen = set()
for t, name, ks, shp, site in out:
    print(f"{name:26s} {str(shp)[:50]:52s} {site:62s} {ks[0][:40]}")
print(len(out), "aten ops with kernels")
