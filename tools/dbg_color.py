import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
from oracle import ref_torch as R
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
col_sd = {k: T(v) for k, v in synth.color_state_dict(21).items()}
cp = R.color_params_from_state_dict(col_sd)
cnet = ops.PackedNet("color", dev)
cnet.set_raw_from_effective([w.to(dev) for w in cp["W"]], [b.to(dev) for b in cp["b"]])
    cnet.pack()
rs = np.random.RandomState(5); n = int(sys.argv[1]); prec = 3
x = T(rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)); d = T(rs.standard_normal((n, 3)).astype(np.float32)); d = d / d.norm(dim=-1, keepdim=True)
normal = T(rs.standard_normal((n, 3)).astype(np.float32)); feat = T((rs.standard_normal((n, 256)) * 0.3).astype(np.float32)); c_rgb = T(rs.standard_normal((n, 3)).astype(np.float32))
cp64 = {"W": [w.double().requires_grad_(True) for w in cp["W"]], "b": [b.double().requires_grad_(True) for b in cp["b"]]}
nrm64 = normal.double().requires_grad_(True); feat64 = feat.double().requires_grad_(True)
rgb_ref, us, zs = R.color_forward(x.double(), nrm64, d.double(), feat64, cp64, keep=True)
for z in zs: z.retain_grad()
(rgb_ref * c_rgb.double()).sum().backward()
cst = ops.ColStash(n, dev, prec)
rgb = ops.color_fwd(cnet.blob, n, prec, normal.to(dev), feat.to(dev), cst, True, pts=x.to(dev).contiguous(), dirs=d.to(dev).contiguous())
d_feat, d_normal = ops.color_bwd(cnet.blob, n, prec, c_rgb.to(dev), rgb, cst)
torch.cuda.synchronize()
for l in range(4):
    u = cst.u.float().sum(0)[l].cpu().double()
    print("u", l, (u - torch.relu(zs[l]).detach()).abs().max().item())
for l in range(5):
    zb = cst.zbar.float().sum(0)[l].cpu().double()
    if l == 4: zb = zb.reshape(-1)[: n * 32].reshape(n, 32)[:, :3]
    ref = zs[l].grad
    print("zbar", l, (zb - ref).abs().max().item(), ref.abs().max().item())
print("d_feat", (d_feat.cpu().double() - feat64.grad).abs().max().item(), feat64.grad.abs().max().item())
err = (d_feat.cpu().double() - feat64.grad).abs().max(dim=1)[0]
bad = torch.nonzero(err > 1e-6).reshape(-1)
print("bad rows", bad[:20].tolist(), len(bad))
