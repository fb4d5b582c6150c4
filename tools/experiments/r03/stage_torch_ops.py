"""Torch-side operators of one eager stage-2 / stage-3 step by GPU time and launch count:
python tools/experiments/r03/stage_torch_ops.py stage2|stage3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from fneus.trainer import synthetic_batches
which = sys.argv[1] if len(sys.argv) > 1 else "stage2"
if which == "stage2":
    from fneus.trainer2 import Stage2Trainer as T
else:
    from fneus.trainer3 import Stage3Trainer as T
dev = torch.device("cuda:0")
tr = T(dev, use_graph=False)
b = synthetic_batches(2, 512, dev)
tr.train_step(b[0]); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr._fixed_shape_step(b[1]); torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 and not e.key.startswith("fneus")]
tot = sum(e.self_device_time_total for e in ev)
print(f"{which}: torch-side self GPU time {tot:.0f} us")
for e in sorted(ev, key=lambda e: -e.self_device_time_total)[:28]:
    print(f"{e.key[:60]:60s} n={e.count:4d} self gpu {e.self_device_time_total:8.1f} us")
