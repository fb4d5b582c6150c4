"""K1 with h6 products (csrc/sdf_h6_kernels.hip) against the shipped two-pass kernel: error vs fp64 and time per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
from oracle import ref_torch as R
dev = torch.device("cuda:0")
for seed, kw in [(20, {}), (3, dict(perturb=0.1)), (5, dict(perturb=0.05, warp=(2, 0.15)))]:
    sd = {k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(seed, **kw).items()}
    p = R.sdf_params_from_state_dict(sd)
    net = ops.PackedNet("sdf", dev)
    net.set_raw_from_effective([w.to(dev) for w in p["W"]], [b.to(dev) for b in p["b"]])
    net.pack()
    for n in (32768, 65536, 1 << 20):
        g = torch.Generator(device=dev).manual_seed(1)
        x = (torch.rand(n, 3, device=dev, generator=g) * 2.2 - 1.1).contiguous()
        sub = x[:8192].cpu().double()
        ref = R.sdf_only(sub, {"W": [w.double() for w in p["W"]], "b": [b.double() for b in p["b"]], "scale": 1.0})[:, 0]
        res = {}
        for name, fn in (("bf3", lambda: ops.sdf_fwd(net.blob, n, 3, pts=x)), ("h6", lambda: ops.sdf_fwd_h6(net.blob, n, pts=x, repack=False)),
                         ("bf1", lambda: ops.sdf_fwd(net.blob, n, 1, pts=x))):
            ops.h6_blob(net.blob)
            out = fn()
            torch.cuda.synchronize()
            err = (out[:8192].cpu().double() - ref).abs().max().item()
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = (err, e0.elapsed_time(e1) / 20 * 1e3)
        print(f"net seed {seed} n {n}: " + "  ".join(f"{k}: err {v[0]:.2e} {v[1]:.1f} us" for k, v in res.items()), flush=True)
