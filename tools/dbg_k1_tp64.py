#!/usr/bin/env python3
"""K1 (large launch) against K2's sdf output on the same points, and its time: run with FNEUS_K1_TP64=0 / 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, synth
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()
for prec in (3, 1):
    for n in ([int(a) for a in sys.argv[1:]] or (32768, 65536, 40003)):
        x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
        st = ops.SdfStash(n, dev, prec, False)
        ref, _, _ = ops.sdf_fwd_grad(net.blob, n, prec, st, False, pts=x)
        out = ops.sdf_fwd(net.blob, n, prec, pts=x)
        same = all(torch.equal(ops.sdf_fwd(net.blob, n, prec, pts=x), out) for _ in range(10))
        for _ in range(3): ops.sdf_fwd(net.blob, n, prec, pts=x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.sdf_fwd(net.blob, n, prec, pts=x)
        e1.record(); torch.cuda.synchronize()
        print(f"prec {prec} n={n}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us   max |K1 - K2| {(out - ref).abs().max().item():.2e}   repeatable {same}")
