#!/usr/bin/env python3
"""Train stage 1 on the analytic two-sphere scene (models/dataset.py) for N steps with the HIP path, then extract the mesh and
score it against the analytic surface (Chamfer-L1, evaluation/chamfer.py).  Usage: train_scene.py [steps] [rays]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np
import torch
from evaluation.chamfer import evaluate_mesh
from fneus.trainer import Stage1Trainer
from models.dataset import SyntheticDataset, scene_surface_points

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
ds = SyntheticDataset(n_images=24, H=192, W=256, device=dev, seed=1)
tr = Stage1Trainer(dev, use_graph=True, lr=5e-4)
gt = scene_surface_points(60000, seed=0)


def chamfer():
    v, f = tr.renderer.extract_geometry([-1.01] * 3, [1.01] * 3, 128)
    return evaluate_mesh(v, f, gt, thresh=0.01, max_dist=1.0)[2] if len(f) else float("nan")


print(f"step      0: Chamfer-L1 {chamfer():.4f} (initial sphere)")
torch.manual_seed(0)
t0 = time.time()
acc = []
for it in range(1, steps + 1):
    # warm-up then cosine decay like exp_runner.py:229-238 (warm_up_end = 10 % of the run here)
    wu = 0.1 * steps
    lr = 5e-4 * (it / wu if it < wu else 0.5 * (1 + np.cos(np.pi * (it - wu) / (steps - wu))) * 0.95 + 0.05)
    tr.set_lr(lr)
    out = tr.train_step(ds.gen_random_rays_at(np.random.RandomState(it).randint(ds.n_images), rays))
    acc.append(out["loss"].detach().clone())
    if it % max(steps // 6, 1) == 0:
        torch.cuda.synchronize()
        recent = torch.stack(acc[-100:]).mean().item()
        print(f"step {it:6d}: loss (mean of last 100) {recent:.4f}  {time.time() - t0:6.1f} s  Chamfer-L1 {chamfer():.4f}")
assert all(torch.isfinite(p).all() for p in tr.params)
