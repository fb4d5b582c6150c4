#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE implementation.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/gen_golden.py

The reference's hot path is imported unmodified with the shims of SURVEY.md section 8(c)
(stub modules for absent third-party imports; numpy.math).  Network weights and rays come from
fneus.synth (numpy RandomState, seeds recorded in each fixture), so the fixtures hold only seeds,
small inputs and the reference's outputs.  Large tensors (parameter gradients) are stored as a
strided subsample + per-tensor L2 norms.
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
REF = "/root/reference"

GRAD_STRIDE = 997


def import_reference():
    for name in ("mcubes", "icecream", "imageio", "cv2"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["icecream"].ic = lambda *a, **k: None
    np.math = math                         # math_utils.py:27,44,52 use np.math.factorial (numpy<2)
    sys.path.insert(0, REF)
    from models import embedder, fields, renderer     # noqa
    return embedder, fields, renderer


def to_t(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def build_nets(fields, seeds):
    from fneus import synth
    torch.manual_seed(0)
    sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5,
                            scale=1.0, geometric_init=True, weight_norm=True)
    col = fields.RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=4,
                                  weight_norm=True, multires_view=4, squeeze_out=True)
    var = fields.SingleVarianceNetwork(init_val=0.3)
    nerf = fields.NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                       use_viewdirs=True)
    ref = fields.RefColor()
    # materialise LazyLinear (fields.py:282, 295-299) with one dummy call
    ref(torch.zeros(2, 3), torch.zeros(2, 256), torch.ones(2, 3), torch.ones(2, 3))
    sdf.load_state_dict(to_t(synth.sdf_state_dict(seeds["sdf"])))
    col.load_state_dict(to_t(synth.color_state_dict(seeds["color"])))
    ref.load_state_dict(to_t(synth.refcolor_state_dict(seeds["refcolor"])))
    nerf.load_state_dict(to_t(synth.nerf_state_dict(seeds["nerf"])))
    return sdf, col, var, nerf, ref


def subsample(t):
    return t.detach().reshape(-1)[::GRAD_STRIDE].numpy().copy()


def gen_units(embedder, fields, renderer, out_dir):
    from fneus import synth
    seeds = {"sdf": 10, "color": 11, "refcolor": 12, "nerf": 13}
    sdf, col, var, nerf, ref = build_nets(fields, seeds)
    rs = np.random.RandomState(100)
    x = torch.from_numpy((rs.uniform(-1.2, 1.2, size=(96, 3))).astype(np.float32))
    x[0] = 0.0                                   # exact zero input
    x[1] = torch.tensor([0.5, -0.25, 0.125])
    d = torch.from_numpy(rs.standard_normal((96, 3)).astype(np.float32))
    d = d / d.norm(dim=-1, keepdim=True)
    res = {"seed_sdf": 10, "seed_color": 11, "seed_refcolor": 12, "seed_nerf": 13, "x": x.numpy(), "dirs": d.numpy()}
    emb6, _ = embedder.get_embedder(6, 3)
    emb4, _ = embedder.get_embedder(4, 3)
    res["embed6"] = emb6(x).numpy()
    res["embed4"] = emb4(d).numpy()
    with torch.no_grad():
        res["sdf_forward"] = sdf(x).numpy()
    xg = x.clone()
    g = sdf.gradient(xg).squeeze(1)
    res["sdf_gradient"] = g.detach().numpy()
    feat = sdf(x)[:, 1:].detach()
    with torch.no_grad():
        res["color"] = col(x, g.detach(), d, feat).numpy()
        rr = ref(x, feat, d, g.detach())
        for k, v in rr.items():
            res["ref_" + k] = v.numpy()
        p4 = torch.from_numpy(rs.uniform(-1, 1, size=(40, 4)).astype(np.float32))
        res["nerf_in"] = p4.numpy()
        a, rgb = nerf(p4, d[:40])
        res["nerf_alpha"], res["nerf_rgb"] = a.numpy(), rgb.numpy()
    # double backward: d/dparam of a random functional of sdf, feature, normal
    c_s = torch.from_numpy(rs.standard_normal((96, 1)).astype(np.float32))
    c_f = torch.from_numpy((rs.standard_normal((96, 256)) * 0.05).astype(np.float32))
    c_n = torch.from_numpy(rs.standard_normal((96, 3)).astype(np.float32))
    sdf.zero_grad()
    xg = x.clone()
    out = sdf(xg)
    gn = sdf.gradient(xg).squeeze(1)
    L = (out[:, :1] * c_s).sum() + (out[:, 1:] * c_f).sum() + (gn * c_n).sum()
    L.backward()
    res["dbl_cs"], res["dbl_cf"], res["dbl_cn"] = c_s.numpy(), c_f.numpy(), c_n.numpy()
    res["dbl_L"] = np.float64(L.item())
    for name, prm in sdf.named_parameters():
        res["dbl_grad_sub/" + name] = subsample(prm.grad)
        res["dbl_grad_norm/" + name] = np.float64(prm.grad.double().norm().item())
    # sampler units
    B, m = 12, 24
    z = np.sort(rs.uniform(0.5, 4.0, size=(B, m)), axis=1).astype(np.float32)
    w = rs.uniform(0, 1, size=(B, m - 1)).astype(np.float32)
    w[0, :] = 0.0                                 # degenerate pdf
    w[1, 3:] = 0.0
    w[2, :] = 1e-9
    res["pdf_bins"], res["pdf_weights"] = z, w
    res["pdf_samples"] = renderer.sample_pdf(torch.from_numpy(z), torch.from_numpy(w), 8, det=True).numpy()
    ro = torch.from_numpy(synth.ray_batch(B, seed=5)[:, :3])
    rd = torch.from_numpy(synth.ray_batch(B, seed=5)[:, 3:6])
    near = -(ro * rd).sum(-1, keepdim=True) - 1.0
    zz = near + 2.0 * torch.linspace(0, 1, m)[None, :]
    rnd = renderer.NeuSRenderer(16, 16, 0, 4, 1.0, sdf_network=sdf)
    with torch.no_grad():
        s0 = sdf.sdf((ro[:, None, :] + rd[:, None, :] * zz[..., None]).reshape(-1, 3)).reshape(B, m)
        res["ups_rays_o"], res["ups_rays_d"], res["ups_z"], res["ups_sdf"] = ro.numpy(), rd.numpy(), zz.numpy(), s0.numpy()
        for inv_s in (64, 512):
            res[f"ups_new_z_{inv_s}"] = rnd.up_sample(ro, rd, zz, s0, 8, inv_s).numpy()
    np.savez_compressed(os.path.join(out_dir, "units.npz"), **res)
    print("units.npz written")


def gen_render(fields, renderer, out_dir, name, B, n_samples, n_importance, n_outside, cos_anneal_ratio,
               ray_seed, n_miss, inside_rays, mask_weight, seeds, white_bkgd=False, adam_steps=3):
    from fneus import synth
    sdf, col, var, nerf, ref = build_nets(fields, seeds)
    data = torch.from_numpy(synth.ray_batch(B, seed=ray_seed, n_miss=n_miss))
    if inside_rays:
        data[:inside_rays, :3] = data[:inside_rays, :3] * 0.2      # ray origins inside the unit sphere
    rays_o, rays_d, true_rgb, mask_in = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
    a = (rays_d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (rays_o * rays_d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    near, far = mid - 1.0, mid + 1.0
    rnd = renderer.NeuSRenderer(n_samples, n_importance, n_outside, 4, 1.0, nerf=nerf, sdf_network=sdf,
                                deviation_network=var, color_network=col, refColor_network=ref)
    trace = []
    orig_cat = rnd.cat_z_vals

    def cat_hook(ro, rd, z, new_z, s, last=False):
        zz, ss = orig_cat(ro, rd, z, new_z, s, last=last)
        trace.append((new_z.clone(), zz.clone(), ss.clone()))
        return zz, ss

    rnd.cat_z_vals = cat_hook
    core = {}
    orig_core = rnd.render_core

    def core_hook(*a_, **k_):
        r = orig_core(*a_, **k_)
        core.update(r)
        return r

    rnd.render_core = core_hook
    params = list(nerf.parameters()) + list(sdf.parameters()) + list(var.parameters()) + \
        list(col.parameters()) + list(ref.parameters())
    opt = torch.optim.Adam(params, lr=5e-4)
    bg = torch.ones([1, 3]) if white_bkgd else None
    res = {"data": data.numpy(), "B": B, "n_samples": n_samples, "n_importance": n_importance,
           "n_outside": n_outside, "cos_anneal_ratio": cos_anneal_ratio, "mask_weight": mask_weight,
           "white_bkgd": int(white_bkgd), "ray_seed": ray_seed,
           **{"seed_" + k: v for k, v in seeds.items()}}
    igr_weight, surface_weight = 0.1, 0.1
    import torch.nn.functional as F
    for step in range(adam_steps):
        trace.clear()
        out = rnd.render(rays_o, rays_d, near, far, perturb_overwrite=0, background_rgb=bg,
                         cos_anneal_ratio=cos_anneal_ratio)
        # exp_runner.py:141-177
        mask = (mask_in > 0.5).float() if mask_weight > 0.0 else torch.ones_like(mask_in)
        mask_sum = mask.sum() + 1e-5
        color_error = (out["color_fine"] - true_rgb) * mask
        color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
        sm = out["sdf_mask"]
        mask_sdf_sum = mask[sm].sum() + 1e-5
        sce = surface_weight * (out["surface_color"][sm] - true_rgb[sm]) * mask[sm]
        surface_color_loss = F.l1_loss(sce, torch.zeros_like(sce), reduction="sum") / mask_sdf_sum
        eik = out["gradient_error"]
        mask_loss = F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
        loss = color_fine_loss + surface_color_loss + eik * igr_weight + mask_loss * mask_weight
        opt.zero_grad()
        loss.backward()
        if step == 0:
            for k, v in out.items():
                res["out/" + k] = v.detach().numpy()
            for k in ("sdf", "dists", "mid_z_vals", "cdf", "s_val"):
                res["core/" + k] = core[k].detach().numpy()
            for i, (nz, zz, ss) in enumerate(trace):
                res[f"trace/new_z_{i}"], res[f"trace/z_{i}"], res[f"trace/sdf_{i}"] = nz.numpy(), zz.numpy(), ss.numpy()
            res["loss/loss"] = np.float64(loss.item())
            res["loss/color"] = np.float64(color_fine_loss.item())
            res["loss/surface"] = np.float64(surface_color_loss.item())
            res["loss/eikonal"] = np.float64(eik.item())
            res["loss/mask"] = np.float64(mask_loss.item())
            for net_name, net in (("sdf", sdf), ("color", col), ("var", var), ("refcolor", ref), ("nerf", nerf)):
                for pname, prm in net.named_parameters():
                    if prm.grad is None:
                        continue
                    res[f"grad_sub/{net_name}.{pname}"] = subsample(prm.grad)
                    res[f"grad_norm/{net_name}.{pname}"] = np.float64(prm.grad.double().norm().item())
        opt.step()
        if step in (0, adam_steps - 1):
            for net_name, net in (("sdf", sdf), ("color", col), ("var", var)):
                for pname, prm in net.named_parameters():
                    res[f"adam{step + 1}_sub/{net_name}.{pname}"] = subsample(prm)
            res[f"adam{step + 1}_loss"] = np.float64(loss.item())
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **res)
    print(name + ".npz written; loss", res["loss/loss"], "n_sdf_mask", int(res["out/sdf_mask"].sum()))


def main():
    torch.set_num_threads(8)
    embedder, fields, renderer = import_reference()
    seeds = {"sdf": 20, "color": 21, "refcolor": 22, "nerf": 23}
    gen_units(embedder, fields, renderer, HERE)
    gen_render(fields, renderer, HERE, "render_wmask_b16_n16", B=16, n_samples=16, n_importance=16, n_outside=0,
               cos_anneal_ratio=1.0, ray_seed=31, n_miss=2, inside_rays=2, mask_weight=0.1, seeds=seeds)
    gen_render(fields, renderer, HERE, "render_wmask_b8_n64", B=8, n_samples=64, n_importance=64, n_outside=0,
               cos_anneal_ratio=1.0, ray_seed=32, n_miss=1, inside_rays=0, mask_weight=0.1, seeds=seeds)
    gen_render(fields, renderer, HERE, "render_womask_b16_n16_o8", B=16, n_samples=16, n_importance=16, n_outside=8,
               cos_anneal_ratio=0.3, ray_seed=33, n_miss=2, inside_rays=1, mask_weight=0.0, seeds=seeds,
               white_bkgd=True)
    gen_render(fields, renderer, HERE, "render_wmask_b16_n16_c0", B=16, n_samples=16, n_importance=16, n_outside=0,
               cos_anneal_ratio=0.0, ray_seed=34, n_miss=0, inside_rays=0, mask_weight=0.1, seeds=seeds,
               adam_steps=1)


if __name__ == "__main__":
    main()
