"""How far each weight gradient of stage 3 is from the reference fixture, as a fraction of the test's scale
(tests/test_hip_stage3.py::test_mateillu_render_vs_reference).  Run once per library build (FNEUS_LIB)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in (ROOT, os.path.join(ROOT, "factored-neus_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_hip_stage3 as t3  # noqa: E402
from fneus.trainer3 import stage3_loss  # noqa: E402

import models.inverRender as IR  # noqa: E402
_orig = IR._seq_direct
_seen = {}


def _spy(seq, x, owner):
    if seq is getattr(owner, "net_cs", None):
        _seen["x"] = x.detach().double().cpu()
    return _orig(seq, x, owner)


IR._seq_direct = _spy
tag = sys.argv[1] if len(sys.argv) > 1 else "a"
for name in ("mateillu_render_b24_n32", "mateillu_render_b128_n64"):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    tr = t3.build(g)
    data, near, far = t3.rays(g)
    out = tr.renderer.mateIllu_render(data[:, :3].contiguous(), data[:, 3:6].contiguous(), near, far,
                                      u_theta=t3.T(g["step0/u_theta"]).to(t3.DEV), u_phi=t3.T(g["step0/u_phi"]).to(t3.DEV))
    mask = (data[:, 9:10] > 0.5).float()
    L = stage3_loss(out, data[:, 6:9], mask)
    L["loss"].backward()
    print(name, "loss", float(L["loss"]), "ref", float(g["step0/loss"]))
    # leaky-ReLU inputs of net_cs (fp64 from the captured input): which sides of the kink, how close to it
    h = _seen["x"]
    pre = []
    for m in tr.mateIllu_network.net_cs:
        if isinstance(m, torch.nn.Linear):
            h = h @ m.weight.detach().double().cpu().t() + m.bias.detach().double().cpu()
            pre.append(h.clone())
        elif isinstance(m, torch.nn.LeakyReLU):
            h = torch.nn.functional.leaky_relu(h, 0.2)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    f = os.path.join(ROOT, "gpurun_out", f"r04_netcs_pre_{name}_{tag}.pt")
    torch.save([p.float() for p in pre[:4]], f)
    other = os.path.join(ROOT, "gpurun_out", f"r04_netcs_pre_{name}_a.pt")
    if tag != "a" and os.path.exists(other):
        for li, (p, q) in enumerate(zip(pre[:4], torch.load(other))):
            flip = (p > 0) != (q.double() > 0)
            rows = flip.any(dim=1).nonzero().reshape(-1).tolist()
            print(f"  net_cs layer {2 * li}: {int(flip.sum())} kink sides differ of {flip.numel()} (rows {rows[:8]}, units "
                  f"{flip.any(dim=0).nonzero().reshape(-1).tolist()[:8]}), |pre| there "
                  f"{[f'{v:.1e}' for v in p[flip].abs().tolist()[:8]]}; max |pre - pre'| {float((p - q.double()).abs().max()):.2e}")
    for k, prm in tr.mateIllu_network.named_parameters():
        ref_sub, ref_norm = g["grad_sub/" + k], float(g["grad_norm/" + k])
        sub = prm.grad.detach().cpu().reshape(-1)[::997].numpy()
        scale = max(ref_norm / np.sqrt(prm.numel()), np.abs(ref_sub).max(), 1e-7)
        e = np.abs(sub - ref_sub)
        print(f"  {k:28s} worst/scale {e.max() / scale:.3e}  at {int(e.argmax())} of {e.size}  norm rel "
              f"{abs(prm.grad.double().norm().item() - ref_norm) / ref_norm:.2e}  scale {scale:.2e}")
