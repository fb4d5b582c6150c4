import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
net = ops.PackedNet("sdf", dev).load_state_dict(sd); net.pack()
def run(n, prec, train, gprec, p2):
    os.environ["FNEUS_K2_P2"] = str(p2)
    torch.manual_seed(3)
    xx = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
    st = ops.SdfStash(n, dev, prec, train, gprec)
    for t in (st.h, st.a, st.feat):
        if t is not None: t.zero_()
    st.ps.zero_()
    out = ops.sdf_fwd_grad(net.blob, n, prec, st, train, pts=xx)
    torch.cuda.synchronize()
    return out, st, xx
for (n, gprec) in ((65536, 3), (40003, 3), (40003, 1), (65536 - 32, 3)):
    (o0, s0, xx), (o1, s1, _) = run(n, 3, True, gprec, 0), run(n, 3, True, gprec, 1)
    (o2, s2, _) = run(n, 3, True, gprec, 1)
    print("n", n, "gprec", gprec, "tiles", s0.tiles, "T", s0.h.shape[2])
    for name in ("h", "feat"):
        a, b, c = getattr(s0, name).float(), getattr(s1, name).float(), getattr(s2, name).float()
        if name == "feat": a, b, c = a[:, None], b[:, None], c[:, None]
        for l in range(a.shape[1]):
            F = 14 if (name == "h" and l == 3) else 16
            d = (a[:, l, :, :F] - b[:, l, :, :F]).abs()
            rep = (c[:, l, :, :F] - b[:, l, :, :F]).abs().max()
            if float(d.max()) > 0:
                idx = torch.nonzero(d > 0)
                print(f"  {name}[{l}] max {float(d.max()):.3e} count {idx.shape[0]} repeat-diff {float(rep):.1e} first {idx[0].tolist()} last {idx[-1].tolist()}  P-values {sorted(set(idx[:,0].tolist()))} tiles {sorted(set(idx[:,1].tolist()))[:6]}.. frags {sorted(set(idx[:,2].tolist()))}")
