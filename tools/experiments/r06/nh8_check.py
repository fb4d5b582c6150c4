"""Round 6: K3 and the colour backward on 256-sample workgroups (FNEUS_R8_NH=8, bf16-activation chains) against 128: planes equal?"""
import os, sys
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
for n in (65536, 70001):
    net = ops.PackedNet("sdf", dev).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(22).items()}); net.pack()
    g = torch.Generator(device=dev).manual_seed(7)
    x = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1).contiguous()
    ds, df, dn = torch.randn(n, device=dev, generator=g), torch.randn(n, 256, device=dev, generator=g) * 0.1, torch.randn(n, 3, device=dev, generator=g)
    st = ops.SdfStash(n, dev, 3, True, 1)
    ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
    res = {}
    for nh in ("4", "8"):
        os.environ["FNEUS_R8_NH"] = nh
        b = ops.SdfBwdBufs(n, dev, 3, 1)
        ops.sdf_bwd(net.blob, n, 3, st, b, ds, df, dn, pts=x); torch.cuda.synchronize()
        res[nh] = b
    print("K3 n", n, {k: bool(torch.equal(getattr(res["4"], k), getattr(res["8"], k))) for k in ("qbar", "adj", "zbar", "zsdf")})
    cnet = ops.PackedNet("color", dev).load_state_dict({k: T(v) for k, v in synth.color_state_dict(23).items()}); cnet.pack()
    d = torch.randn(n, 3, device=dev, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    nrm, feat, c_rgb = torch.randn(n, 3, device=dev, generator=g), (torch.randn(n, 256, device=dev, generator=g) * 0.3).contiguous(), torch.randn(n, 3, device=dev, generator=g)
    cs = ops.ColStash(n, dev, 3, gprec=2)
    rgb = ops.color_fwd(cnet.blob, n, 3, nrm, feat, cs, True, pts=x, dirs=d)
    out = {}
    for nh in ("4", "8"):
        os.environ["FNEUS_R8_NH"] = nh
        cs.zbar.zero_()
        dfe, dno = ops.color_bwd(cnet.blob, n, 3, c_rgb, rgb, cs); torch.cuda.synchronize()
        out[nh] = (dfe.clone(), dno.clone(), cs.zbar.clone())
    print("colour bwd n", n, [bool(torch.equal(a, b)) for a, b in zip(out["4"], out["8"])])
os.environ.pop("FNEUS_R8_NH")
