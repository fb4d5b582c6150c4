"""Standalone timing of K2 (train) and K3 at N = 65 536 in parity mode (timing variants: FNEUS_LIB=...)."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
net = ops.PackedNet("sdf", dev).load_state_dict(sd); net.pack()
n = 65536
xx = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
prec = 3
st_t = ops.SdfStash(n, dev, prec, True)
bufs = ops.SdfBwdBufs(n, dev, prec)
ds, df, dn = torch.randn(n, device=dev), torch.randn(n, 256, device=dev), torch.randn(n, 3, device=dev)
k2 = timeit(lambda: ops.sdf_fwd_grad(net.blob, n, prec, st_t, True, pts=xx))
k3 = timeit(lambda: ops.sdf_bwd(net.blob, n, prec, st_t, bufs, ds, df, dn, pts=xx))
print(f"{os.environ.get('FNEUS_LIB', 'default'):60s} K2 train {k2:7.1f} us   K3 {k3:7.1f} us")
