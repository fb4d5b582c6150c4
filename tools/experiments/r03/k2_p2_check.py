"""K2 as two launches (p2 forward with the stash + reverse sweep) against the fused 32-sample kernel: outputs, stash planes
and time.  FNEUS_K2_P2 is read at every call."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
net = ops.PackedNet("sdf", dev).load_state_dict(sd); net.pack()
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
def run(n, prec, train, gprec, p2):
    os.environ["FNEUS_K2_P2"] = str(p2)
    torch.manual_seed(3)
    xx = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
    st = ops.SdfStash(n, dev, prec, train, gprec)
    for t in (st.h, st.a, st.feat):
        if t is not None: t.zero_()
    st.ps.zero_()
    out = ops.sdf_fwd_grad(net.blob, n, prec, st, train, pts=xx)
    torch.cuda.synchronize()
    return out, st, xx
for (n, prec, train, gprec) in ((65536, 3, True, 1), (40003, 3, True, 3), (40003, 3, False, 1), (65536, 1, True, 1), (32737 + 64, 1, False, 1)):
    (o0, s0, xx), (o1, s1, _) = run(n, prec, train, gprec, 0), run(n, prec, train, gprec, 1)
    msg = [f"n={n} prec={prec} train={train} gprec={gprec}:"]
    for name, a, b in zip(("sdf", "feat", "normal"), o0, o1):
        msg.append(f"{name} {float((a - b).abs().max()):.2e}")
    tl = s0.tiles
    for l in range(8):
        d = (s0.sigma(l) - s1.sigma(l)).abs().max()
        if float(d) > 4e-5: msg.append(f"sigma[{l}] {float(d):.2e}")
    if train:
        for name in ("pe", "h", "feat", "a"):
            a, b = getattr(s0, name), getattr(s1, name)
            if name in ("h", "a"):
                for l in range(8):
                    F = 14 if (name == "h" and l == 3) or (name == "a" and l == 3) else 16
                    d = (a[:, l, :, :F].float() - b[:, l, :, :F].float()).abs().max()
                    nz = (b[:, l, tl:].float().abs().max()) if b.shape[2] > tl else 0.0
                    if float(d) > 0 or float(nz) > 0: msg.append(f"{name}[{l}] {float(d):.2e} pad {float(nz):.1e}")
            else:
                d = (a.float() - b.float()).abs().max()
                if float(d) > 0: msg.append(f"{name} {float(d):.2e}")
    print(" ".join(msg), flush=True)
n = 65536
xx = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
for prec, train in ((3, True), (3, False)):
    st = ops.SdfStash(n, dev, prec, train)
    for p2 in (0, 1, 2, 3):
        os.environ["FNEUS_K2_P2"] = str(p2)
        print(f"prec {prec} train {train} p2={p2}: {timeit(lambda: ops.sdf_fwd_grad(net.blob, n, prec, st, train, pts=xx)):.1f} us")
