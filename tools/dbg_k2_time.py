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
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for prec in (3, 1):
    st_t = ops.SdfStash(n, dev, prec, True); st_f = ops.SdfStash(n, dev, prec, False)
    print("prec", prec, "K1", timeit(lambda: ops.sdf_fwd(net.blob, n, prec, pts=xx)))
    print("prec", prec, "K2 train", timeit(lambda: ops.sdf_fwd_grad(net.blob, n, prec, st_t, True, pts=xx)))
    print("prec", prec, "K2 infer", timeit(lambda: ops.sdf_fwd_grad(net.blob, n, prec, st_f, False, pts=xx)))
    bufs = ops.SdfBwdBufs(n, dev, prec)
    ds, df, dn = torch.randn(n, device=dev), torch.randn(n, 256, device=dev), torch.randn(n, 3, device=dev)
    print("prec", prec, "K3", timeit(lambda: ops.sdf_bwd(net.blob, n, prec, st_t, bufs, ds, df, dn, pts=xx)))
# small launches of K1 (the 16-new-samples evaluations of the hierarchical sampler: 8192 points)
for prec in (3, 1):
    for n_small in (8192, 32768):
        xs = xx[:n_small].contiguous()
        print("prec", prec, f"K1 n={n_small}", timeit(lambda: ops.sdf_fwd(net.blob, n_small, prec, pts=xs)))
