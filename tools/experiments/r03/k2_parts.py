"""K2 (sdf_fwd_grad) standalone at 65 536 points, parity mode: train / inference launch.  Run with FNEUS_LIB=<variant> to
time the kernel with parts compiled out (FNEUS_DBG_K2_NO_REVERSE, FNEUS_DBG_NO_PLANESTORE)."""
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
prec = 3
st_t = ops.SdfStash(n, dev, prec, True); st_f = ops.SdfStash(n, dev, prec, False)
print(os.environ.get("FNEUS_LIB", "base"), "K1 %.1f" % timeit(lambda: ops.sdf_fwd(net.blob, n, prec, pts=xx)),
      "K2 train %.1f" % timeit(lambda: ops.sdf_fwd_grad(net.blob, n, prec, st_t, True, pts=xx)),
      "K2 infer %.1f" % timeit(lambda: ops.sdf_fwd_grad(net.blob, n, prec, st_f, False, pts=xx)))
