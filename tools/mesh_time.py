#!/usr/bin/env python3
"""Time validate_mesh's compute at the reference's resolution (512^3 grid through K1 + surface extraction) on the
synthetic (perturbed-sphere) SDF network.  Usage: mesh_time.py [resolution]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np
import torch
from fneus.trainer import Stage1Trainer

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
tr = Stage1Trainer(dev)
bmin, bmax = [-1.01] * 3, [1.01] * 3
for r in (64, res):
    torch.cuda.synchronize(); t0 = time.time()
    u = tr.renderer.extract_sdf_grid(bmin, bmax, r)
    torch.cuda.synchronize(); t1 = time.time()
    from models.mesh import marching_tetrahedra
    v, f = marching_tetrahedra(u, 0.0)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"resolution {r}: SDF grid {r**3 / 1e6:.1f} M points in {t1 - t0:.3f} s ({r**3 / (t1 - t0) / 1e6:.0f} M points/s), "
          f"surface {len(v)} vertices / {len(f)} triangles in {t2 - t1:.3f} s")
with torch.no_grad():
    s = tr.sdf_network.sdf((v / (res - 1.0) * 2.02 - 1.01).contiguous())
print(f"max |sdf| at the mesh vertices: {s.abs().max().item():.2e}")
