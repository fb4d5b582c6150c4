import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/factored-neus_amd")
import numpy as np, torch
from fneus import ops, synth, pp
dev = torch.device("cuda:0"); n = 65536
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(22).items()}); net.pack()
g = torch.Generator(device=dev).manual_seed(7)
x = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1).contiguous()
ds, df, dn = torch.randn(n, device=dev, generator=g), torch.randn(n, 256, device=dev, generator=g) * 0.1, torch.randn(n, 3, device=dev, generator=g)
st = ops.SdfStash(n, dev, 3, True, 2); ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
w16 = ops.h16_blob(net, 0)
def run(mode):
    b = ops.SdfBwdBufs(n, dev, 3, 2)
    b.zbar[0, 8].copy_(pp.pack(df, 16, 1)[0])
    b.c.w16 = w16.data_ptr() if mode == "h16" else None
    ops.sdf_bwd(net.blob, n, 3, st, b, ds, None, dn, pts=x); torch.cuda.synchronize(); return b
r, t = run("xhi"), run("h16")
# c planes: [P, T, 8, 16, 64, 8] lane-private; compare per (tile, layer) ratio
cr = r.cs[0].float()                       # bf16 values
ct = t.cs[0].view(torch.float16).float()   # fp16 values (scaled)
for l in (7, 3, 0):
    a, b = cr[:, l].reshape(cr.shape[0], -1), ct[:, l].reshape(ct.shape[0], -1)
    # per tile: least-squares ratio (should be the tile's... per-sample scale varies; take medians of elementwise ratio where |a| large)
    ratio = (b / a)[a.abs() > a.abs().mean()]
    lr = torch.log2(ratio.abs())
    print("layer", l, "c16 / c: log2 ratio median %.2f, frac non-integer exponent %.3f, sign agreement %.4f" % (lr.median().item(), ((lr - lr.round()).abs() > 0.02).float().mean().item(), (ratio > 0).float().mean().item()))
for l in (7, 0):
    v0, v1 = pp.value(r.zbar[:, l], n), pp.value(t.zbar[:, l], n)
    num = (v1 * v0).sum(1); den = (v0 * v0).sum(1)
    k = (num / den)
    print("zbar", l, "per-sample projection v1.v0/v0.v0: median %.3f, 10%% %.3f 90%% %.3f" % (k.median().item(), k.quantile(0.1).item(), k.quantile(0.9).item()))
