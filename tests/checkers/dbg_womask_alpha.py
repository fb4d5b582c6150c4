#!/usr/bin/env python3
"""womask: dL/d(raw density) of the background NeRF per sample, HIP path vs the CPU restatement (teacher-forced depths)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import test_hip_render as H
import test_oracle_golden as O
from oracle import ref_torch as R
name = sys.argv[1] if len(sys.argv) > 1 else "render_womask_b64_n64_o32"
g = H.load(os.path.join(ROOT, "tests", "golden"), name)
keep = []
orig = R.nerf_forward
def hook(*a, **k):
    alpha, rgb = orig(*a, **k)
    alpha.retain_grad(); rgb.retain_grad(); keep.append((alpha, rgb))
    return alpha, rgb
R.nerf_forward = hook
out_o, losses, trace, leaves = O.run_oracle_render(g, requires_grad=True, teacher_z=True)
losses["loss"].backward()
R.nerf_forward = orig
ga_o, gr_o = keep[-1][0].grad.reshape(-1), keep[-1][1].grad.reshape(-1, 3)
for fused in (True,):
    kept = []
    rnd, nets = H.build(g, 3, 3)
    f0 = nets["nerf"].forward

    def fwd(*a, **k):
        r = f0(*a, **k)
        for t in r[:2]:
            t.retain_grad()
        kept.append(r)
        return r

    nets["nerf"].forward = fwd
    data = H.T(g["data"]).to(H.DEV)
    rays_o, rays_d, rgb, mask = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
    near, far = R.near_far_from_sphere(rays_o, rays_d)
    out = rnd.render(rays_o, rays_d, near, far, perturb_overwrite=0, cos_anneal_ratio=float(g["cos_anneal_ratio"]),
                     background_rgb=None, z_vals_override=H.final_z(g).to(H.DEV),
                     loss_args=(rgb, mask, 0.1, float(g["mask_weight"]), 0.1))
    loss = out["losses"]["loss"]
    loss.backward()
    a, c = kept[-1][0], kept[-1][1]
    ga, gr = a.grad.detach().cpu().reshape(-1), c.grad.detach().cpu().reshape(-1, 3)
    print(f"fused={fused}: alpha fwd max diff {(a.detach().cpu().reshape(-1) - keep[-1][0].detach().reshape(-1)).abs().max():.2e}")
    print(f"  d alpha: sum hip {ga.sum():.6e} oracle {ga_o.sum():.6e}; max|diff| {(ga - ga_o).abs().max():.3e} of max {ga_o.abs().max():.3e}; "
          f"sum|diff| {(ga - ga_o).abs().sum():.3e}; signed sum diff {(ga - ga_o).sum():.3e}")
    print(f"  d rgb:   max|diff| {(gr - gr_o).abs().max():.3e} of max {gr_o.abs().max():.3e}")
    d = (ga - ga_o)
    i = d.abs().argmax().item()
    print("  worst sample", i, "ray", i // int(g["n_outside"]), "k", i % int(g["n_outside"]), float(ga[i]), float(ga_o[i]))
    r = i // 160
    sl = slice(r * 160 + 150, r * 160 + 160)
    print("  raw density hip", a.detach().cpu().reshape(-1)[sl].numpy())
    print("  raw density ora", keep[-1][0].detach().reshape(-1)[sl].numpy())
    print("  grad hip", ga[sl].numpy())
    print("  grad ora", ga_o[sl].numpy())
    top = d.abs().topk(8).indices
    print("  top:", [(int(j) // int(g["n_outside"]), int(j) % int(g["n_outside"]), f"{float(d[j]):.2e}") for j in top])
