import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
from oracle import ref_torch as R
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}
p = R.sdf_params_from_state_dict(sd)
net = ops.PackedNet("sdf", dev)
net.set_raw_from_effective([w.to(dev) for w in p["W"]], [b.to(dev) for b in p["b"]])
net.pack()
n = 128
x = (torch.rand(n, 3, device=dev) * 2.2 - 1.1).contiguous()
out = torch.zeros(8192, device=dev)
ops.sdf_fwd_h6(net.blob, n, pts=x, out=out)
torch.cuda.synchronize()
print("sdf nan per tile:", [int(torch.isnan(out[32 * t:32 * t + 32]).sum()) for t in range(4)])
d = out[256:256 + 8 * 512].cpu().numpy().view(np.float16).reshape(8, 2, 256 * 2)
for l in range(1, 8):
    for w, t in ((0, 0), (1, 2)):
        v = d[l, w].astype(np.float32)
        print("behind layer", l, "tile", t, "nan", int(np.isnan(v).sum()), "inf", int(np.isinf(v).sum()), "max", float(np.nanmax(np.abs(v))), "first", v[:4])
