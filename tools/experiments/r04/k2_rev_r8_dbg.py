"""where do the r8 reverse sweep and the 4-wave one differ: per layer max |a_l| difference (decoded planes), normals"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(21).items()}); net.pack()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40003
x = (torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 2 - 1).contiguous()
def run(r8, gprec=3):
    os.environ["FNEUS_K2_REV8"] = str(r8)
    st = ops.SdfStash(n, dev, 3, True, gprec)
    st.a.zero_()
    out = ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
    torch.cuda.synchronize()
    return out, st
for rep in range(2):
    (s0, f0, n0), st0 = run(0)
    (s1, f1, n1), st1 = run(1)
    (s2, f2, n2), st2 = run(1)
    print("normal r8 vs 4w:", (n1 - n0).abs().max().item(), " r8 vs r8:", (n2 - n1).abs().max().item(), "rows differing", ((n1 - n0).abs().amax(1) > 0).sum().item(), "of", n)
    bad = ((n1 - n0).abs().amax(1) > 0).nonzero().flatten()
    print(" first differing samples:", bad[:16].tolist(), " mod 64:", (bad[:16] % 64).tolist())
    for l in range(7, -1, -1):
        a1, a0 = st1.plane(st1.a, l), st0.plane(st0.a, l)
        d = (a1 - a0).abs()
        print(" a_%d: max diff %.3e (max |a| %.3e)  planes equal: %s  rows differing %d" % (l, d.max().item(), a0.abs().max().item(), torch.equal(st1.a[:, l], st0.a[:, l]), (d.reshape(d.shape[0], -1).amax(1) > 0).sum().item() if d.dim() == 2 else -1))
