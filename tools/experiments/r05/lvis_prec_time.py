#!/usr/bin/env python3
"""fneus_lvis_visibility at the stage-3 step's size (512 points, 128 lobes x 32 directions): time and error of the one-product modes
against the three-product parity mode"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
from models.fields import Lvis
from models.inverRender import visibility_sample_dirs

dev = torch.device("cuda:0")
net = Lvis()
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.lvis_state_dict(5).items()})
net.to(dev)
g = torch.Generator().manual_seed(1)
n = 512
pts = ((torch.rand(n, 3, generator=g) - 0.5) * 1.2).to(dev)
nrm = torch.randn(n, 3, generator=g)
nrm = (nrm / nrm.norm(dim=-1, keepdim=True)).to(dev)
sg = torch.from_numpy(np.asarray(synth.mateillu_state_dict(32)["lgtSGs"])).to(dev)
dirs, w = ops.vis_sample_dirs_sgs(sg.contiguous(), torch.rand(128, 32, generator=g).to(dev), torch.rand(128, 32, generator=g).to(dev))
res = {}
modes = [("parity (3 bf16 products)", ops.PREC_PARITY), ("one bf16 product", ops.PREC_FAST)]
if hasattr(ops, "PREC_H16"):
    modes.append(("one fp16 product", ops.PREC_H16))
for name, prec in modes:
    net.set_precision(prec)
    for _ in range(3):
        v = net.visibility(pts, nrm, dirs, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        v = net.visibility(pts, nrm, dirs, w)
    e1.record()
    torch.cuda.synchronize()
    res[name] = v.clone()
    err = (v - res["parity (3 bf16 products)"]).abs().max().item()
    print(f"{name:28s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us   max |difference to parity| of a lobe's visibility {err:.2e}")
