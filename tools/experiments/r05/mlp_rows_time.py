#!/usr/bin/env python3
"""time per launch of the fneus_mlp_* kernels at the stage-2 / 3 shapes, against torch (rocBLAS) for the same product"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch

from fneus import ops

dev = torch.device("cuda:0")
torch.backends.cuda.preferred_blas_library("cublas")


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, n_in, n_out in [(512, 512, 512), (512, 63, 512), (512, 512, 32), (512, 256, 256), (2048, 256, 256), (2048, 90, 256),
                          (512, 128, 4), (2048, 256, 1)]:
    x = torch.randn(rows, n_in, device=dev)
    w = torch.randn(n_out, n_in, device=dev) * 0.05
    b = torch.randn(n_out, device=dev)
    y = torch.empty(rows, n_out, device=dev)
    dy = torch.randn(rows, n_out, device=dev)
    dx = torch.empty(rows, n_in, device=dev)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    job = dict(x=x, weight=w, bias=b, y=y, dy=dy, dx=dx, d_weight=dw, d_bias=db, rows=rows, n_in=n_in, n_out=n_out, act=0, act_in=1)
    fwd = dict(job, act=1)
    t_f = timeit(lambda: ops.mlp_forward([fwd]))
    t_x = timeit(lambda: ops.mlp_backward_input([job]))
    t_w = timeit(lambda: ops.mlp_backward_params([job]))
    t_lf = timeit(lambda: torch.relu_(torch.addmm(b, x, w.t(), out=y)))
    t_lx = timeit(lambda: torch.mm(dy, w, out=dx))
    t_lw = timeit(lambda: (torch.mm(dy.t(), x, out=dw), torch.sum(dy, 0, out=db)))
    print(f"rows {rows:5d} {n_in:4d} -> {n_out:4d}: forward {t_f:6.2f} us (torch addmm + relu {t_lf:6.2f}), input grad {t_x:6.2f} ({t_lx:6.2f}), "
          f"params {t_w:6.2f} ({t_lw:6.2f})")
