"""Standalone timing of fneus_dw_gemm_pp on random planes with the SDF network's job mix (N = 65 536)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, pp

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
gprec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = pp.n_tiles(n)
P = 2 if gprec == 3 else 1
mk = lambda L, F: (torch.randn(P, L, T, F, 64, 8, device=dev) * 0.5).bfloat16()
zbar, h, a, adj = mk(9, 16), mk(8, 16), mk(8, 16), mk(8, 16)
pe, qb = mk(1, 4), mk(1, 4)
grad = torch.zeros(9 * 65536 + 4096, dtype=torch.float32, device=dev)
for target in (int(x) for x in os.environ.get("WGS", "256,192,128,512").split(",")):
    jobs = ops.GemmPPJobs(dev, "sdf", target_wgs=target)
    O = ops.PPOperand
    base = grad.data_ptr()
    for l in range(1, 8):
        jobs.add(O(zbar[:, l], 0, 8), O(h[:, l - 1], 0, 8), base + 4 * 65536 * l, 256, 256, 256,
                 A2=O(a[:, l], 0, 8), B2=O(adj[:, l - 1], 0, 8), bias_ptr=base + 4 * 9 * 65536 + 4 * 256 * l)
    jobs.add(O(zbar[:, 8], 0, 8), O(h[:, 7], 0, 8), base + 4 * 65536 * 8, 256, 256, 256)
    jobs.add(O(zbar[:, 0], 0, 8), O(pe[:, 0], 0, 2), base, 39, 256, 39, A2=O(a[:, 0], 0, 8), B2=O(qb[:, 0], 0, 2))
    jobs.add(O(zbar[:, 4], 0, 8), O(pe[:, 0], 0, 2), base + 4 * 20000, 39, 256, 39, A2=O(a[:, 4], 0, 8), B2=O(qb[:, 0], 0, 2))
    jobs.finalize(T)
    byts = sum(b for b in jobs.bytes) * T * P
    for _ in range(3):
        jobs.run(gprec=gprec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        jobs.run(gprec=gprec)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"gemm_pp gprec={gprec} n={n} wgs={jobs.n_wgs}: {ms*1000:.1f} us, {byts/1e9:.3f} GB -> {byts/ms/1e9:.2f} TB/s")
