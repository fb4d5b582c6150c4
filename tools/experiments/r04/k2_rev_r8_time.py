"""K2 at 65 536 points: forward launch alone (FNEUS_K2_P2=2), reverse sweep alone (3) and both (1), with the reverse sweep on the
4-wave kernel (FNEUS_K2_REV8=0) and on resident-weight 8-wave workgroups (1).  FNEUS_LIB=<variant> for builds with parts compiled out."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
net = ops.PackedNet("sdf", dev).load_state_dict(sd); net.pack()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
xx = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = ops.SdfStash(n, dev, 3, True)
os.environ["FNEUS_K2_P2"] = "1"
ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=xx)       # sigma' blocks in place
for r8 in (0, 1):
    os.environ["FNEUS_K2_REV8"] = str(r8)
    res = []
    for p2 in (3, 1):
        os.environ["FNEUS_K2_P2"] = str(p2)
        res.append("P2=%d: %.1f us" % (p2, timeit(lambda: ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=xx))))
    print(os.environ.get("FNEUS_LIB", "base").split("/")[-1], "REV8=%d" % r8, "  ".join(res), flush=True)
