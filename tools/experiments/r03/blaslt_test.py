import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import torch
which = sys.argv[1]
if which != "default":
    torch.backends.cuda.preferred_blas_library(which)
print("blas:", torch.backends.cuda.preferred_blas_library())
dev = torch.device("cuda:0")
a = torch.randn(256, 512, device=dev); b = torch.randn(512, 256, device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("mm [256,512]x[512,256]: %.1f us" % t(lambda: a @ b))
print("mm via (b.T @ a.T).T: %.1f us" % t(lambda: (b.t() @ a.t()).t()))
c = torch.randn(512, 90, device=dev)
print("mm [256,512]x[512,90]: %.1f us" % t(lambda: a @ c))
x = torch.randn(512, 256, device=dev); dy = torch.randn(512, 256, device=dev)
print("dW = dy.T @ x: %.1f us" % t(lambda: dy.t() @ x))
from fneus.trainer import synthetic_batches
from fneus.trainer3 import Stage3Trainer
from fneus.trainer2 import Stage2Trainer
for T in (Stage2Trainer, Stage3Trainer):
    tr = T(dev, use_graph=True)
    bs = synthetic_batches(4, 512, dev)
    for i in range(4): tr.train_step(bs[i])
    torch.cuda.synchronize()
    print(T.__name__, "%.3f ms" % (t(lambda: tr.train_step(bs[0]), 20) / 1e3))
