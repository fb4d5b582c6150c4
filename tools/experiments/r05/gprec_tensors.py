#!/usr/bin/env python3
"""Which parameter tensors' gradients need the lo planes?  tests/test_hip_render.py::test_loss_and_gradients' comparison against the
reference's gradients, per tensor, at gradient precision 1 and 3: the tensors that pass the gprec-3 bounds (5e-3 of scale per
sampled element, 2e-3 of the norm) only with hi + lo planes"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import test_hip_render as T

gd = os.path.join(ROOT, "tests", "golden")
for name in T.WMASK[:2] + T.BIG:
    g = T.load(gd, name)
    rows = {}
    for gprec in (1, 3):
        out, nets, (rgb, mask) = T.run(g, 3, teacher_z=True, fused_loss=True, gprec=gprec)
        out["losses"]["loss"].backward()
        for key in g:
            if not key.startswith("grad_norm/"):
                continue
            pname = key[len("grad_norm/"):]
            net, rest = pname.split(".", 1)
            if nets.get(net) is None:
                continue
            prm = dict(nets[net].named_parameters())[rest]
            ref_norm = float(g[key])
            ref_sub = g["grad_sub/" + pname]
            sub = prm.grad.detach().cpu().reshape(-1)[::997].numpy()
            scale = max(ref_norm / np.sqrt(prm.numel()), np.abs(ref_sub).max(), 1e-7)
            rows.setdefault(pname, {})[gprec] = (np.abs(sub - ref_sub).max() / scale, abs(prm.grad.double().norm().item() - ref_norm) / (ref_norm + 1e-12))
    print(f"== {name}")
    for pname, r in sorted(rows.items()):
        (s1, n1), (s3, n3) = r[1], r[3]
        flag = "  <-- needs lo" if (s1 > 5e-3 or n1 > 2e-3) and not pname.startswith(("refcolor", "nerf")) else ""
        if s1 > 2e-3 or n1 > 1e-3 or flag:
            print(f"  {pname:44s} gprec1 sub {s1:.2e} norm {n1:.2e} | gprec3 sub {s3:.2e} norm {n3:.2e}{flag}")
