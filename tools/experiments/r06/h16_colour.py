"""[runs against the prototype at commit beb36f6 only: fneus_h16_pack and the w16 fields were reverted, DESIGN.md 4.1f]
Round 6: the colour backward with ONE fp16 product per multiplication (per-sample scaled cotangents, FneusColStash.w16) against the
hi + lo chain and the bf16-cotangent chain: outputs and planes, at three cotangent magnitudes; time."""
import os, sys
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth, pp
dev = torch.device("cuda:0"); n = 65536
net = ops.PackedNet("color", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.color_state_dict(23).items()}); net.pack()
g = torch.Generator(device=dev).manual_seed(9)
x = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1).contiguous()
d = torch.randn(n, 3, device=dev, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
nrm, feat = torch.randn(n, 3, device=dev, generator=g), (torch.randn(n, 256, device=dev, generator=g) * 0.3).contiguous()
c0 = torch.randn(n, 3, device=dev, generator=g)
# per-sample magnitudes over 12 orders (compositing weights do that)
mag = torch.exp(torch.empty(n, 1, device=dev).uniform_(-14.0, 0.0, generator=g) * 2.0)
st = ops.ColStash(n, dev, 3, gprec=2)
rgb = ops.color_fwd(net.blob, n, 3, nrm, feat, st, True, pts=x, dirs=d)
w16 = ops.h16_blob(net, 1)
def run(mode, c):
    os.environ["FNEUS_COLB_XHI"] = "0" if mode == "hilo" else "1"
    st.zbar.zero_()
    a, b = ops.color_bwd(net.blob, n, 3, c, rgb, st, w16=w16 if mode == "h16" else None); torch.cuda.synchronize()
    return a.clone(), b.clone(), st.zbar.clone()
for name, c in (("unit", c0), ("x 1e-9", c0 * 1e-9), ("x 1e6", c0 * 1e6), ("12 orders per sample", c0 * mag)):
    ref = run("hilo", c.contiguous())
    for mode in ("xhi", "h16"):
        out = run(mode, c.contiguous())
        row = []
        for i, nm in ((0, "d_feat"), (1, "d_normal")):
            row.append("%s L2 %.2e max %.2e" % (nm, (out[i] - ref[i]).norm().item() / ref[i].norm().item(), (out[i] - ref[i]).abs().max().item() / ref[i].abs().max().item()))
        zl = [(pp.value(out[2][:, l], n) - pp.value(ref[2][:, l], n)).norm().item() / pp.value(ref[2][:, l], n).norm().item() for l in range(4)]
        # per-sample relative error of d_feat: the scale must not matter
        ps = ((out[0] - ref[0]).norm(dim=1) / (ref[0].norm(dim=1) + 1e-38))
        print(f"{name:22s} {mode}: " + "; ".join(row) + "; zbar planes L2 " + " ".join("%.1e" % v for v in zl) + "; per-sample d_feat error: median %.1e max %.1e, finite %s" % (ps.median().item(), ps.max().item(), bool(torch.isfinite(out[0]).all())))
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
plane = torch.zeros(2 * ((n + 63) // 64), 16, 64, 8, dtype=torch.bfloat16, device=dev)
os.environ["FNEUS_COLB_XHI"] = "1"
print("time, fragments out: bf16 cotangents %.1f us, one fp16 product %.1f us (+ fp16 copy of the weights %.1f us)" % (
    timeit(lambda: ops.color_bwd(net.blob, n, 3, c0, rgb, st, dfeat_plane=plane)),
    timeit(lambda: ops.color_bwd(net.blob, n, 3, c0, rgb, st, dfeat_plane=plane, w16=w16)), timeit(lambda: ops.h16_blob(net, 1))))
