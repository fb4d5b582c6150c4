"""a few launches of K1 (parity mode) at 1 M points, for rocprofv3 --pmc passes and for timing: python k1_one.py [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, synth
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()})
net.pack()
n = 1 << 20
x = (torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 2.2 - 1.1).contiguous()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for _ in range(2):
    ops.sdf_fwd(net.blob, n, 3, pts=x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.sdf_fwd(net.blob, n, 3, pts=x)
e1.record()
torch.cuda.synchronize()
print(f"FNEUS_K1_W8_BIG={os.environ.get('FNEUS_K1_W8_BIG', '31')}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch of {n} points")
