#!/usr/bin/env python3
"""Time of a full-resolution validation render (1600 x 1200 rays, forward only) in 512-ray and 4096-ray chunks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus.trainer import Stage1Trainer
from models.dataset import SyntheticDataset
dev = torch.device("cuda:0")
tr = Stage1Trainer(dev)
ds = SyntheticDataset(n_images=1, H=1200, W=1600, device=dev)
o, d = ds.gen_rays_at(0, 1)
o, d = o.reshape(-1, 3), d.reshape(-1, 3)
for chunk in (512, 4096, 16384):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        outs = []
        for oo, dd in zip(o.split(chunk), d.split(chunk)):
            data = torch.cat([oo, dd, torch.zeros(len(oo), 4, device=dev)], -1)
            outs.append(tr.render_only(data)["color_fine"].clone())
        img = torch.cat(outs).cpu()
        dt = time.time() - t0
    print(f"chunk {chunk:6d} rays: {dt:.2f} s per 1600x1200 image ({o.shape[0] * 128 / dt / 1e7:.2f}e7 ray-samples/s)")
