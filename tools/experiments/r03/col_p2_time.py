"""Colour-network forward at 65 536 samples: two-pass kernel (FNEUS_COL_P2=1) against the 4-wave kernels, training / inference."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
net = ops.PackedNet("color", dev).load_state_dict({k: T(v) for k, v in synth.color_state_dict(2).items()}); net.pack()
n = 65536
g = torch.Generator(device=dev).manual_seed(1)
pts = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1).contiguous()
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1).contiguous()
nrm = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1).contiguous()
feat = torch.randn(n, 256, device=dev, generator=g).contiguous()
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
allo = {}
for train in (True, False):
    st = ops.ColStash(n, dev, 3) if train else None
    outs = {}
    for p2 in (0, 1):
        os.environ["FNEUS_COL_P2"] = str(p2)
        f = lambda: ops.color_fwd(net.blob, n, 3, nrm, feat, st, train, pts=pts, dirs=dirs)
        outs[p2] = f().clone()
        print(f"train={train} p2={p2}: {timeit(f):.1f} us")
    print("   max |rgb difference|", float((outs[0] - outs[1]).abs().max()))
    allo[train] = outs
    os.environ["FNEUS_COL_P2"] = "1"
    again = ops.color_fwd(net.blob, n, 3, nrm, feat, st, train, pts=pts, dirs=dirs)
    print("   p2 repeat difference", float((again - outs[1]).abs().max()))
for a in (0, 1):
    for b in (0, 1):
        d = (allo[True][a] - allo[False][b]).abs()
        print(f"train p2={a} vs infer p2={b}: {float(d.max()):.2e}  rows differing > 1e-5: {int((d.max(dim=1).values > 1e-5).sum())}, first {torch.nonzero(d.max(dim=1).values > 1e-5)[:6].flatten().tolist()}")
