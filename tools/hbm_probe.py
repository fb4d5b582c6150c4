#!/usr/bin/env python3
"""What streaming bandwidth does this MI355X actually deliver to simple kernels?  (calibration for the roofline notes)"""
import time
import torch

dev = torch.device("cuda:0")
n = 1 << 29                       # 2 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


gb = n * 4 / 1e9
print(f"read  (sum):      {gb / t(lambda: a.sum()):8.0f} GB/s")
print(f"write (fill):     {gb / t(lambda: b.fill_(1.0)):8.0f} GB/s")
print(f"copy  (r+w):      {2 * gb / t(lambda: b.copy_(a)):8.0f} GB/s")
print(f"axpy  (2r+w):     {3 * gb / t(lambda: torch.add(a, b, out=b)):8.0f} GB/s")
h = a.view(torch.bfloat16)
print(f"read bf16 (sum):  {gb / t(lambda: h.sum(dtype=torch.float32)):8.0f} GB/s")
