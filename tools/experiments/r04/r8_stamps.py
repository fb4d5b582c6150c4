"""phase cycles of the r8 reverse sweep (FNEUS_LIB=.../libfneus_r8_stamps.so): one launch at 65 536 points, printf from block 0"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()
n = 65536
xx = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
st = ops.SdfStash(n, dev, 3, True)
os.environ["FNEUS_K2_P2"] = "1"
ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=xx)
torch.cuda.synchronize()
os.environ["FNEUS_K2_P2"] = "3"
for _ in range(3):
    ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=xx)
    torch.cuda.synchronize()
# K3 (sdf_bwd_r8_kernel) on the same stash
bufs = ops.SdfBwdBufs(n, dev, 3, 1)
ds, df, dn = torch.randn(n, device=dev), torch.randn(n, 256, device=dev), torch.randn(n, 3, device=dev)
for _ in range(3):
    ops.sdf_bwd(net.blob, n, 3, st, bufs, ds, df, dn, pts=xx)
    torch.cuda.synchronize()
