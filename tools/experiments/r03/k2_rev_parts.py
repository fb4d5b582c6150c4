"""The reverse sweep of K2 as its own launch (FNEUS_K2_P2=3) at 65 536 points; with FNEUS_LIB=<variant> parts compiled out."""
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
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
st = ops.SdfStash(n, dev, 3, True)
res = []
for p2 in (2, 3, 1):
    os.environ["FNEUS_K2_P2"] = str(p2)
    res.append("%d: %.1f" % (p2, timeit(lambda: ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=xx))))
print(os.environ.get("FNEUS_LIB", "base").split("/")[-1], "  ".join(res))
