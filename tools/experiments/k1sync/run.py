import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
net = ops.PackedNet("sdf", dev).load_state_dict(sd); net.pack()
n = 65536
xx = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
ref = ops.sdf_fwd(net.blob, n, 3, pts=xx)
here = os.path.dirname(os.path.abspath(__file__))
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("baseline K1 x3", timeit(lambda: ops.sdf_fwd(net.blob, n, 3, pts=xx)), "x1", timeit(lambda: ops.sdf_fwd(net.blob, n, 1, pts=xx)))
for S in (0, 1, 2, 4, 8):
    lib = C.CDLL(os.path.join(here, f"libk1sync_{S}.so"))
    out = torch.empty(n, device=dev)
    for prec in (3, 1):
        f = lambda: lib.k1sync_run(C.c_void_p(net.blob.data_ptr()), C.c_void_p(xx.data_ptr()), C.c_long(n), C.c_void_p(out.data_ptr()), prec, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        t = timeit(f)
        err = (out - ref).abs().max().item() if prec == 3 else -1
        print(f"sync every {S} stages prec {prec}: {t:8.1f} us  err vs baseline {err:.2e}")
