import os, sys, torch
sys.path.insert(0, "factored-neus_amd")
from fneus.optim import FlatAdam
dev = torch.device("cuda:0")
ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (65536 * 9, 65536 * 4 + 74000, 257, 256, 3000, 512 * 257)]
opt = FlatAdam(ps, lr=1e-3) if True else None
for p in ps: p.grad = torch.randn_like(p)
for _ in range(5): opt.step()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): opt.step()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"adam: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per step for {sum(p.numel() for p in ps)} values")
