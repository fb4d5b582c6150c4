#!/usr/bin/env python3
"""Kernel times of K4 (colour network forward / backward) at N = 65 536, parity and bf16 mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sd = {k: torch.from_numpy(v) for k, v in synth.color_state_dict(21).items()}
net = ops.PackedNet("color", dev).load_state_dict(sd); net.pack()
x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1).contiguous()
nrm = torch.randn(n, 3, device=dev); feat = (torch.randn(n, 256, device=dev) * 0.3).contiguous(); c = torch.randn(n, 3, device=dev)

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for prec in (3, 1):
    st = ops.ColStash(n, dev, prec)
    rgb = ops.color_fwd(net.blob, n, prec, nrm, feat, st, True, pts=x, dirs=d)
    print(f"prec {prec}: fwd infer {timeit(lambda: ops.color_fwd(net.blob, n, prec, nrm, feat, None, False, pts=x, dirs=d)):.0f} us  "
          f"fwd train {timeit(lambda: ops.color_fwd(net.blob, n, prec, nrm, feat, st, True, pts=x, dirs=d)):.0f} us  "
          f"bwd {timeit(lambda: ops.color_bwd(net.blob, n, prec, c, rgb, st)):.0f} us")
