import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
from oracle import ref_torch as R
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
p = R.sdf_params_from_state_dict(sd)
net16 = ops.PackedNet("sdf16", dev); net16.set_raw_from_effective([w.to(dev) for w in p["W"]], [b.to(dev) for b in p["b"]]); net16.pack()
net32 = ops.PackedNet("sdf", dev); net32.set_raw_from_effective([w.to(dev) for w in p["W"]], [b.to(dev) for b in p["b"]]); net32.pack()
rs = np.random.RandomState(7)
x = T(rs.uniform(-1.1, 1.1, size=(1000, 3)).astype(np.float32)); xd = x.to(dev).contiguous()
p64 = {"W": [w.double() for w in p["W"]], "b": [b.double() for b in p["b"]], "scale": 1.0}
ref = R.sdf_only(x.double(), p64)[:, 0]
for prec in (3, 1):
    o16 = ops.sdf_fwd16(net16.blob, 1000, prec, pts=xd)
    o32 = ops.sdf_fwd(net32.blob, 1000, prec, pts=xd)
    print("prec", prec, "err16", (o16.cpu().double() - ref).abs().max().item(), "err32", (o32.cpu().double() - ref).abs().max().item())
for n in (8192, 32768, 65536, 262144):
    xx = torch.rand(n, 3, device=dev) * 2 - 1
    for prec in (3, 1):
        for name, fn, blob in (("e32", ops.sdf_fwd, net32.blob), ("e16", ops.sdf_fwd16, net16.blob)):
            for _ in range(3): fn(blob, n, prec, pts=xx)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): fn(blob, n, prec, pts=xx)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print(f"n={n:7d} prec={prec} {name}: {dt*1e6:8.1f} us  {n/dt/1e6:8.1f} Mpts/s  {n*1.049e6*(3 if prec==3 else 1)/dt/1e12:7.1f} TF/s mfma-rate")
