#!/usr/bin/env python3
"""Which operand formats keep the stage-3 light visibility within parity?  CPU, fp64 emulation (tests/checkers/num_schemes.py's
products) of get_diffuse_visibility's network part (inverRender.py:163-190): Lvis = sigmoid(MLP 90 -> 256 x 4 -> 1, ReLU) at the 32
directions of each lobe, the facing ones averaged with their weights.  A lobe's visibility averages up to 32 sigmoid outputs
(slope <= 1/4): it forgives more than the SDF chain does."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests", "checkers")); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd")); sys.path.insert(0, ROOT)
import num_schemes as NS
from fneus import synth
torch.set_default_dtype(torch.float64)


def embed(x, L):
    out = [x]
    for k in range(L):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def lvis(sd, pts, dirs, scheme):
    x = torch.cat([embed(pts, 10), embed(dirs, 4)], -1)
    for i, l in enumerate((0, 2, 4, 6, 8)):
        W, b = sd[f"lvis.{l}.weight"].double(), sd[f"lvis.{l}.bias"].double()
        x = (NS.prod(W, x, scheme) if scheme != "fp64" else x @ W.t()) + b
        if i < 4:
            x = torch.relu(x)
    return torch.sigmoid(x)[:, 0]


for seed, gain in ((5, 1.0), (6, 1.0), (7, 1.0), (5, 1.5), (5, 2.0), (5, 3.0)):
    # gain > 1: every hidden layer's weights scaled up -- a sharper network, like a trained one with hard shadow edges
    sd = {k: torch.from_numpy(np.asarray(v)) * (gain if k.endswith("weight") and not k.startswith("lvis.0") else 1.0)
          for k, v in synth.lvis_state_dict(seed).items()}
    g = torch.Generator().manual_seed(seed)
    n, M, S = 24, 128, 32
    pts = (torch.rand(n, 3, generator=g) - 0.5) * 1.2
    dirs = torch.randn(M, S, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    w = torch.rand(M, S, generator=g)
    nrm = torch.randn(n, 3, generator=g)
    nrm = nrm / nrm.norm(dim=-1, keepdim=True)
    front = (torch.einsum("nc,msc->nms", nrm, dirs) > 1e-6).double()
    P = pts[:, None, None, :].expand(n, M, S, 3).reshape(-1, 3)
    D = dirs[None].expand(n, M, S, 3).reshape(-1, 3)
    res = {}
    for scheme in ("fp64", "bf3", "h2a", "h2w", "h1", "bf1"):
        v = lvis(sd, P, D, scheme).reshape(n, M, S) * front
        res[scheme] = (v * w[None]).sum(-1) / (w.sum(-1)[None] + 1e-6)
    ref = res["fp64"]
    print(f"seed {seed} gain {gain}: visibility in [{ref.min():.3f}, {ref.max():.3f}]; max |error| of a lobe's visibility: " +
          ", ".join(f"{s} {float((res[s] - ref).abs().max()):.2e}" for s in ("bf3", "h2a", "h2w", "h1", "bf1")))
