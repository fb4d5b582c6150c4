import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/factored-neus_amd")
import numpy as np, torch
from fneus import ops, synth, pp
T = lambda a: torch.from_numpy(np.asarray(a)); DEV = torch.device("cuda:0")
n = 40003
net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(22).items()}); net.pack()
g = torch.Generator(device=DEV).manual_seed(7)
x = (torch.rand(n, 3, device=DEV, generator=g) * 2 - 1).contiguous()
ds, df, dn = torch.randn(n, device=DEV, generator=g), torch.randn(n, 256, device=DEV, generator=g) * 0.1, torch.randn(n, 3, device=DEV, generator=g)
st = ops.SdfStash(n, DEV, 3, True, 1); ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
def k3(xhi):
    os.environ["FNEUS_BWD_XHI"] = str(xhi)
    b = ops.SdfBwdBufs(n, DEV, 3, 1); ops.sdf_bwd(net.blob, n, 3, st, b, ds, df, dn, pts=x); torch.cuda.synchronize(); return b
r, t = k3(0), k3(1)
for name, slots in (("adj", 8), ("zbar", 9)):
    for l in range(slots):
        v0, v1 = pp.value(getattr(r, name)[:, l], n), pp.value(getattr(t, name)[:, l], n)
        print(name, l, "max/scale %.2e  L2 %.2e" % ((v1 - v0).abs().max().item() / v0.abs().max().item(), (v1 - v0).norm().item() / v0.norm().item()))
cnet = ops.PackedNet("color", DEV).load_state_dict({k: T(v) for k, v in synth.color_state_dict(23).items()}); cnet.pack()
d = torch.randn(n, 3, device=DEV, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
nrm, feat, c_rgb = torch.randn(n, 3, device=DEV, generator=g), (torch.randn(n, 256, device=DEV, generator=g) * 0.3).contiguous(), torch.randn(n, 3, device=DEV, generator=g)
cs = ops.ColStash(n, DEV, 3, gprec=1); rgb = ops.color_fwd(cnet.blob, n, 3, nrm, feat, cs, True, pts=x, dirs=d)
def cb(xhi):
    os.environ["FNEUS_COLB_XHI"] = str(xhi); cs.zbar.zero_()
    a, b = ops.color_bwd(cnet.blob, n, 3, c_rgb, rgb, cs); torch.cuda.synchronize(); return a.clone(), b.clone(), cs.zbar.clone()
r, t = cb(0), cb(1)
for i, nm in ((0, "d_feat"), (1, "d_normal")):
    print(nm, "max/scale %.2e  L2 %.2e" % ((t[i] - r[i]).abs().max().item() / r[i].abs().max().item(), (t[i] - r[i]).norm().item() / r[i].norm().item()))
for l in range(4):
    v0, v1 = pp.value(r[2][:, l], n), pp.value(t[2][:, l], n)
    print("col zbar", l, "max/scale %.2e  L2 %.2e" % ((v1 - v0).abs().max().item() / v0.abs().max().item(), (v1 - v0).norm().item() / v0.norm().item()))
