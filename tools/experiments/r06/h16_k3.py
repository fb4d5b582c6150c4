"""[runs against the prototype at commit beb36f6 only: fneus_h16_pack and the w16 fields were reverted, DESIGN.md 4.1f]
Round 6: K3 with ONE fp16 product per multiplication (per-sample scaled cotangents, FneusSdfBwdBufs.w16, seed as fragments) against the
hi + lo chains and the bf16-cotangent chains: every plane the GEMM reads, at several cotangent magnitudes; time."""
import os, sys
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth, pp
dev = torch.device("cuda:0"); n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(22).items()}); net.pack()
g = torch.Generator(device=dev).manual_seed(7)
x = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1).contiguous()
ds0, df0, dn0 = torch.randn(n, device=dev, generator=g), torch.randn(n, 256, device=dev, generator=g) * 0.1, torch.randn(n, 3, device=dev, generator=g)
mag = torch.exp(torch.empty(n, 1, device=dev).uniform_(-14.0, 0.0, generator=g) * 2.0)
st = ops.SdfStash(n, dev, 3, True, 2); ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
w16 = ops.h16_blob(net, 0)
T_al = 2 * ((n + 63) // 64)
def run(mode, ds, df, dn):
    os.environ["FNEUS_BWD_XHI"] = "0" if mode == "hilo" else "1"
    b = ops.SdfBwdBufs(n, dev, 3, 2)
    if mode == "hilo":
        ops.sdf_bwd(net.blob, n, 3, st, b, ds, df, dn, pts=x)
    else:
        b.zbar[0, 8].copy_(pp.pack(df, 16, 1)[0])
        b.c.w16 = w16.data_ptr() if mode == "h16" else None
        ops.sdf_bwd(net.blob, n, 3, st, b, ds, None, dn, pts=x)
    torch.cuda.synchronize(); return b
cases = (("unit", 1.0, 1.0, 1.0), ("all x 1e-9", 1e-9, 1e-9, 1e-9), ("all x 1e5", 1e5, 1e5, 1e5), ("d_feat x 1e-6 only", 1.0, 1e-6, 1.0), ("d_normal x 1e-6 only", 1.0, 1.0, 1e-6))
for name, a, bq, c in cases + (("12 orders per sample", None, None, None),):
    if a is None: ds, df, dn = (ds0 * mag[:, 0]).contiguous(), (df0 * mag).contiguous(), (dn0 * mag).contiguous()
    else: ds, df, dn = (ds0 * a).contiguous(), (df0 * bq).contiguous(), (dn0 * c).contiguous()
    ref = run("hilo", ds, df, dn)
    for mode in ("xhi", "h16"):
        out = run(mode, ds, df, dn)
        res = []
        for nm, slots in (("adj", 8), ("zbar", 8)):
            errs = []
            for l in range(slots):
                v0, v1 = pp.value(getattr(ref, nm)[:, l], n), pp.value(getattr(out, nm)[:, l], n)
                errs.append((v1 - v0).norm().item() / max(v0.norm().item(), 1e-38))
            res.append(nm + " L2 " + " ".join("%.1e" % e for e in errs))
        fin = all(bool(torch.isfinite(getattr(out, k).float()).all()) for k in ("adj", "zbar", "qbar", "zsdf"))
        same_q = bool(torch.equal(out.qbar, ref.qbar)) and bool(torch.equal(out.zsdf, ref.zsdf))
        print(f"{name:22s} {mode}: " + "; ".join(res) + f"; finite {fin}; qbar / zsdf planes equal {same_q}")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
os.environ["FNEUS_BWD_XHI"] = "1"
b = ops.SdfBwdBufs(n, dev, 3, 2); b.zbar[0, 8].copy_(pp.pack(df0, 16, 1)[0])
t1 = timeit(lambda: ops.sdf_bwd(net.blob, n, 3, st, b, ds0, None, dn0, pts=x))
b.c.w16 = w16.data_ptr()
t2 = timeit(lambda: ops.sdf_bwd(net.blob, n, 3, st, b, ds0, None, dn0, pts=x))
print("K3, seed as fragments: bf16 cotangents %.1f us, one fp16 product %.1f us (+ fp16 copy of the weights %.1f us)" % (t1, t2, timeit(lambda: ops.h16_blob(net, 0))))
