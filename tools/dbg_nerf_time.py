#!/usr/bin/env python3
"""Kernel times of K7 (background NeRF++) at the womask shape: 512 rays x 160 samples = 81 920 points.
Usage: dbg_nerf_time.py [n_points]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np
import torch
from fneus import ops, synth
from models.fields import NeRF

n = int(sys.argv[1]) if len(sys.argv) > 1 else 81920
dev = torch.device("cuda:0")
F_NERF = 2 * 604160
for prec in (ops.PREC_PARITY, ops.PREC_FAST):
    net = NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(3).items()})
    net.to(dev)
    net.set_precision(prec)
    rs = np.random.RandomState(0)
    p = rs.standard_normal((n, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True)
    pts4 = torch.from_numpy(np.concatenate([p, rs.uniform(0.02, 1, (n, 1))], 1).astype(np.float32)).to(dev)
    d = rs.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    dirs = torch.from_numpy(d.astype(np.float32)).to(dev)
    net.refresh()
    be = net._be
    stash = ops.NerfStash(n, dev, prec)
    dd, dr = torch.randn(n, device=dev), torch.randn(n, 3, device=dev)
    jobs = ops.nerf_dw_jobs(be.net, stash, n)

    def timeit(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_inf = timeit(lambda: ops.nerf_fwd(be.net.blob, n, prec, pts4, dirs, None, False))
    t_fwd = timeit(lambda: ops.nerf_fwd(be.net.blob, n, prec, pts4, dirs, stash, True))
    t_bwd = timeit(lambda: ops.nerf_bwd(be.net.blob, n, prec, dd, dr, stash))
    t_dw = timeit(lambda: jobs.run(n, prec))
    tf = lambda ms: n * F_NERF / (ms * 1e-3) / 1e12
    print(f"prec={prec} n={n}: fwd(no stash) {t_inf * 1e3:.0f} us ({tf(t_inf):.0f} TFLOP/s)  fwd(train) {t_fwd * 1e3:.0f} us  "
          f"bwd {t_bwd * 1e3:.0f} us ({tf(t_bwd):.0f} TFLOP/s)  dW GEMM {t_dw * 1e3:.0f} us ({tf(t_dw):.0f} TFLOP/s)")
    # the same network as stock PyTorch modules (rocBLAS GEMMs), forward + backward
    lin = [torch.nn.Linear(i, o).to(dev) for i, o in zip([84, 256, 256, 256, 256, 340, 256, 256], [256] * 8)]
    heads = [torch.nn.Linear(256, 257).to(dev), torch.nn.Linear(283, 128).to(dev), torch.nn.Linear(128, 3).to(dev)]
    pe, ve = torch.randn(n, 84, device=dev), torch.randn(n, 27, device=dev)

    def torch_step():
        h = pe
        for i, l in enumerate(lin):
            h = torch.relu(l(h))
            if i == 4:
                h = torch.cat([pe, h], -1)
        o = heads[0](h)
        hv = torch.relu(heads[1](torch.cat([o[:, :256], ve], -1)))
        out = heads[2](hv)
        (out.sum() + o[:, 256].sum()).backward()
    if prec == ops.PREC_PARITY:
        print(f"   stock PyTorch fp32 (rocBLAS) fwd + bwd of the same layers, encodings precomputed: {timeit(torch_step, 5) * 1e3:.0f} us")
