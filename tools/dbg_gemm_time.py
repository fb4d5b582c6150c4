#!/usr/bin/env python3
"""Time of the SDF weight-gradient GEMM launch (fneus_dw_gemm, 9 layers x 2 products) at N = 65 536 on random planes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, synth
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()
for prec in (3, 1):
    stash, bufs = ops.SdfStash(n, dev, prec, train=True), ops.SdfBwdBufs(n, dev, prec)
    for t in (stash.pe, stash.h, stash.a, stash.feat, bufs.zbar, bufs.adj):
        t.copy_(torch.randn(t.shape, device=dev).to(t.dtype) * 0.1)
    grad = torch.zeros(net.n_params, dtype=torch.float32, device=dev)
    jobs = ops.sdf_dw_jobs(net, stash, bufs, grad, n)
    for _ in range(3): jobs.run(n, prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): jobs.run(n, prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byts = sum(t.numel() * 2 for t in (stash.h, stash.a, bufs.zbar, bufs.adj))
    print(f"prec {prec}: dw_gemm sdf {ms * 1e3:.0f} us, {len(jobs.jobs)} products, {jobs.tiles} tiles, operand planes {byts / 1e9:.2f} GB -> {byts / ms / 1e9:.2f} TB/s")
