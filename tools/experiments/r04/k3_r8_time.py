"""K3 at 65 536 points: the 4-wave kernel (FNEUS_K3_R8=0) against resident-weight 8-wave workgroups (1); K2 beside it."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
gprec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
xx = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = ops.SdfStash(n, dev, 3, True, gprec)
bufs = ops.SdfBwdBufs(n, dev, 3, gprec)
ds, df, dn = torch.randn(n, device=dev), torch.randn(n, 256, device=dev), torch.randn(n, 3, device=dev)
ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=xx)
for r8 in (0, 1, 0, 1):
    os.environ["FNEUS_K3_R8"] = str(r8)
    print(os.environ.get("FNEUS_LIB", "base").split("/")[-1], "gprec", gprec, "K3_R8=%d: %.1f us" % (r8, timeit(lambda: ops.sdf_bwd(net.blob, n, 3, st, bufs, ds, df, dn, pts=xx))), flush=True)
