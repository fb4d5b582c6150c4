"""one launch of the two-pass K1 at 1 M points with FNEUS_P2_STAMPS builds: where a pass spends its cycles (waves 0 and 4 of block 0)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, synth
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()})
net.pack()
n = 1 << 20
x = (torch.rand(n, 3, device=dev) * 2.2 - 1.1).contiguous()
for _ in range(2):
    ops.sdf_fwd(net.blob, n, 3, pts=x)
    torch.cuda.synchronize()
print(f"(1 M points: 32 units per CU, 18 passes per unit of which 2 have 3 k-steps; per-pass figures = totals / {32 * 18})", flush=True)
