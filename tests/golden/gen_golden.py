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
REF = "/root/reference"

GRAD_STRIDE = 997


def _load_by_path(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# fneus.synth (numpy streams only) is loaded BY PATH: the repo's package directory must never be on sys.path here -- it
# holds a regular package called `models`, which would shadow the reference's namespace package of the same name.
synth = _load_by_path("fneus_synth_for_goldens", os.path.join(ROOT, "factored-neus_amd", "fneus", "synth.py"))


def import_reference(with_dataset=False, with_stage23=False):
    for p in list(sys.path):
        if os.path.abspath(p or ".").startswith(os.path.join(ROOT, "factored-neus_amd")):
            sys.path.remove(p)
    assert "models" not in sys.modules or sys.modules["models"].__path__._path[0].startswith(REF), "repo package imported first"
    for name in ("mcubes", "icecream", "imageio", "cv2", "tifffile", "trimesh"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["icecream"].ic = lambda *a, **k: None
    np.math = math                         # math_utils.py:27,44,52 use np.math.factorial (numpy<2)
    sys.path.insert(0, REF)
    from models import embedder, fields, renderer     # noqa
    for m in (embedder, fields, renderer):
        assert m.__file__.startswith(REF + "/"), f"{m.__name__} was not loaded from the reference: {m.__file__}"
    out = [embedder, fields, renderer]
    if with_dataset:
        # models/dataset.py imports models.rend_util (imageio freeimage plugin download at import, rend_util.py:4): not on
        # the path of the ray generator -- an empty module stands in for it
        if "models.rend_util" not in sys.modules:
            sys.modules["models.rend_util"] = types.ModuleType("models.rend_util")
        from models import dataset
        assert dataset.__file__.startswith(REF + "/")
        out.append(dataset)
    return out


def to_t(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def build_nets(fields, seeds):
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
               ray_seed, n_miss, inside_rays, mask_weight, seeds, white_bkgd=False, adam_steps=3, ray_stride=1,
               sampler_trace=True):
    """ray_stride > 1 (large batches): per-SAMPLE arrays ([B, n] and [B*n, ...]) are stored for rays 0, stride, 2 stride, ...
    only; per-ray arrays, losses and gradients always cover the whole batch."""
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
        trace.append((new_z.clone(), zz.clone(), ss.clone(), z.clone(), s.clone()))
        return zz, ss

    rnd.cat_z_vals = cat_hook
    # the inverse-CDF lookups of sample_pdf (renderer.py:64-66): record cdf, u and the index torch.searchsorted returned
    lookups = []
    orig_ss = torch.searchsorted

    def ss_hook(cdf, u, right=False, **kw):
        r = orig_ss(cdf, u, right=right, **kw)
        lookups.append((cdf.clone(), r.clone()))
        return r
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
    sel = slice(None, None, ray_stride)
    res["ray_stride"] = ray_stride

    def per_sample(t, n_cols=None):
        """[B, n, ...] or [B*n, ...] array restricted to the stored rays"""
        a = t.detach().numpy()
        if ray_stride == 1:
            return a
        if a.shape[0] == B:
            return a[sel]
        return a.reshape((B, -1) + a.shape[1:])[sel].reshape((-1,) + a.shape[1:])

    for step in range(adam_steps):
        trace.clear()
        lookups.clear()
        torch.searchsorted = ss_hook
        try:
            out = rnd.render(rays_o, rays_d, near, far, perturb_overwrite=0, background_rgb=bg,
                             cos_anneal_ratio=cos_anneal_ratio)
        finally:
            torch.searchsorted = orig_ss
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
                big = v.dim() >= 2 and v.shape[0] == B and v.shape[1] not in (1, 3)
                res["out/" + k] = per_sample(v) if big else v.detach().numpy()
            for k in ("sdf", "dists", "mid_z_vals", "cdf", "s_val"):
                res["core/" + k] = per_sample(core[k])
            for i, (nz, zz, ss, z_in, s_in) in enumerate(trace):
                res[f"trace/new_z_{i}"], res[f"trace/z_{i}"], res[f"trace/sdf_{i}"] = per_sample(nz), per_sample(zz), per_sample(ss)
                if sampler_trace:     # inputs of up_sample step i, its cdf and the bin each new depth was drawn from
                    cdf, inds = lookups[i]
                    res[f"trace/z_in_{i}"], res[f"trace/sdf_in_{i}"] = per_sample(z_in), per_sample(s_in)
                    res[f"trace/cdf_{i}"] = per_sample(cdf)
                    res[f"trace/bin_{i}"] = per_sample(torch.clamp(inds - 1, min=0)).astype(np.int16)
            assert not sampler_trace or len(lookups) == len(trace)
            if ray_stride > 1:       # the final depths of EVERY ray: teacher-forced runs of the whole batch need them
                res["trace/z_final"] = trace[-1][1].numpy()
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


def gen_lvis_util(fields, renderer, out_dir, name, B, n_samples, n_importance, ray_seed, seeds):
    """NeuSRenderer.lvis_mateIllu_render_util (renderer.py:503-564): the stage-2/3 entry -- unperturbed hierarchical
    sampling, SDF at the section mid-points, per-ray inside-sphere mask"""
    sdf, col, var, nerf, ref = build_nets(fields, seeds)
    data = torch.from_numpy(synth.ray_batch(B, seed=ray_seed, n_miss=2))
    data[:2, :3] = data[:2, :3] * 0.2                   # two ray origins inside the unit sphere
    rays_o, rays_d = data[:, :3], data[:, 3:6]
    a = (rays_d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (rays_o * rays_d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    near, far = mid - 1.0, mid + 1.0
    rnd = renderer.NeuSRenderer(n_samples, n_importance, 0, 4, 1.0, nerf=nerf, sdf_network=sdf, deviation_network=var,
                                color_network=col, refColor_network=ref)
    with torch.no_grad():
        out = rnd.lvis_mateIllu_render_util(rays_o, rays_d, near, far)
    res = {"data": data.numpy(), "B": B, "n_samples": n_samples, "n_importance": n_importance, "ray_seed": ray_seed,
           **{"seed_" + k: v for k, v in seeds.items()},
           "out/n_samples": np.int64(out["n_samples"]), "out/mid_z_vals": out["mid_z_vals"].numpy(),
           "out/sdf": out["sdf"].numpy(), "out/inside_sphere_mask": out["inside_sphere_mask"].numpy()}
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **res)
    print(name + ".npz written; inside_sphere_mask", int(res["out/inside_sphere_mask"].sum()), "of", B)


def stage2_reference():
    """the stage-2 modules of the reference with the `.cuda()` shim of SURVEY.md section 8(c) (calLvis.py:305, 351-352 move
    fresh CPU tensors to the GPU by hand; there is none here)"""
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models import calLvis
    assert calLvis.__file__.startswith(REF + "/")
    return calLvis



def record_primary_z(rnd, trace):
    """(round 3) wrap NeuSRenderer.cat_z_vals of this renderer object so that the final depths of the primary rays (the call with
    last=True, renderer.py:543-548) land in trace["prim_z"]: stage-2 / stage-3 tests can feed them in (teacher forcing)"""
    real_cat = rnd.cat_z_vals

    def cat_hook(ro, rd, z, new_z, s, last=False):
        z2, s2 = real_cat(ro, rd, z, new_z, s, last=last)
        if last:
            trace["prim_z"] = z2.detach().clone()
        return z2, s2

    rnd.cat_z_vals = cat_hook
    return real_cat


def gen_lvis_render(fields, renderer, out_dir, name, B, n_samples, n_importance, ray_seed, seeds, room, bias=0.5, warp=None,
                    adam_steps=3, outside_rays=0):
    """NeuSRenderer.lvis_render (renderer.py:567-627) -> cal_indiLgt (calLvis.py:339-409), the stage-2 loss of
    lvis.py:164-170, its gradients on Lvis + IndirectLight and `adam_steps` optimiser steps (lvis.py:89-92, 172-174).
    room=True: SDF positive INSIDE a sphere and the camera inside it (every secondary ray can hit the opposite wall);
    room=False: the convex ball of the stage-1 fixtures (secondary rays leave without a hit)."""
    calLvis = stage2_reference()
    sdf, col, var, nerf, ref = build_nets(fields, seeds)
    sdf.load_state_dict(to_t(synth.sdf_state_dict(seeds["sdf"], bias=bias, warp=warp, inside_out=room)))
    lvis, indi = fields.Lvis(), fields.IndirectLight()
    lvis(torch.zeros(2, 3), torch.ones(2, 3))                   # materialise the LazyLinear layers
    indi(torch.zeros(2, 3))
    lvis.load_state_dict(to_t(synth.lvis_state_dict(seeds["lvis"])))
    indi.load_state_dict(to_t(synth.indilgt_state_dict(seeds["indilgt"])))
    data = torch.from_numpy(synth.ray_batch(B, seed=ray_seed, n_miss=0 if room else 2))
    if room:
        data[: B - outside_rays, :3] *= 0.08                    # camera inside the room (the last rays stay outside: no hit)
    rays_o, rays_d = data[:, :3], data[:, 3:6]
    a = (rays_d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (rays_o * rays_d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    near, far = mid - 1.0, mid + 1.0
    if room:
        near = torch.where(near < 0.02, torch.full_like(near, 0.02), near)     # start inside the room, in front of the camera
    rnd = renderer.NeuSRenderer(n_samples, n_importance, 0, 4, 1.0, nerf=nerf, sdf_network=sdf, deviation_network=var,
                                color_network=col, refColor_network=ref, lvis_network=lvis, indiLgt_network=indi)
    params = list(lvis.parameters()) + list(indi.parameters())
    names = ["lvis." + k for k, _ in lvis.named_parameters()] + ["indi." + k for k, _ in indi.named_parameters()]
    opt = torch.optim.Adam(params, lr=5e-4)                     # lvis.py:92 with train.learning_rate of confs/wmask.conf:23
    res = {"data": data.numpy(), "near": near.numpy(), "far": far.numpy(), "B": B, "n_samples": n_samples,
           "n_importance": n_importance, "ray_seed": ray_seed, "room": int(room), "bias": bias,
           "warp": np.asarray(warp if warp is not None else (-1, 0.0), dtype=np.float64), "lr": 5e-4,
           **{"seed_" + k: v for k, v in seeds.items()}}
    glob = renderer.cal_indiLgt.__globals__
    real = {k: glob[k] for k in ("up_sample", "cal_firHit_rgb", "compute_weight", "sample_dirs")}
    real_rand = torch.rand
    for step in range(adam_steps):
        trace, draws = {}, []

        def rand_hook(*a_, **k_):
            r = real_rand(*a_, **k_)
            draws.append(r.clone())
            return r

        def up_hook(*a_, **k_):
            z = real["up_sample"](*a_, **k_)
            trace["z_fine"] = z.clone()
            trace["sec_origins"] = (k_["rays_o"] if "rays_o" in k_ else a_[0]).detach().clone()      # (round 3) the secondary rays' origins: teacher-forced stage-2 test
            trace["inv_s"] = torch.as_tensor(k_["inv_s"]).clone()
            return z

        def hit_hook(*a_, **k_):
            rgb, m = real["cal_firHit_rgb"](*a_, **k_)
            trace["sec_hit_rgb"], trace["sec_sdf_mask"] = rgb.clone(), m.clone()
            return rgb, m

        def w_hook(*a_, **k_):
            w, wi = real["compute_weight"](*a_, **k_)
            trace["sec_weights"], trace["sec_weights_inside"] = w.clone(), wi.clone()
            return w, wi

        def dirs_hook(normals, r_theta, r_phi):
            d = real["sample_dirs"](normals, r_theta, r_phi)
            trace["normal"], trace["dirs"] = normals[:, 0, :].clone(), d.clone()
            return d

        glob.update(up_sample=up_hook, cal_firHit_rgb=hit_hook, compute_weight=w_hook, sample_dirs=dirs_hook)
        torch.rand = rand_hook
        torch.manual_seed(1000 + step)
        real_cat = record_primary_z(rnd, trace)
        try:
            out = rnd.lvis_render(rays_o, rays_d, near, far)
        finally:
            torch.rand = real_rand
            glob.update(real)
            rnd.cat_z_vals = real_cat
        sdf_mask = out["sdf_mask"]
        # lvis.py:164-170
        lvis_error = out["gt_lvis"] - out["pre_lvis"]
        lvis_loss = lvis_error.abs().sum() / (sdf_mask[..., None].expand(out["gt_lvis"].shape).sum() + 1e-6)
        tr_error = (out["gt_trace_radiance"] - out["pre_trace_radiance"]) * sdf_mask[..., None, None]
        tr_loss = tr_error.abs().sum() / (sdf_mask[..., None, None].expand(out["gt_trace_radiance"].shape).sum() + 1e-6)
        loss = lvis_loss + tr_loss
        opt.zero_grad()
        loss.backward()
        tag = f"step{step}/"
        assert len(draws) == 2 and draws[0].shape == (int(sdf_mask.sum()), 4)
        res.update({tag + "u_theta": draws[0].numpy(), tag + "u_z": draws[1].numpy(), tag + "loss": loss.item(),
                    tag + "lvis_loss": lvis_loss.item(), tag + "trace_radiance_loss": tr_loss.item()})
        if step == 0:
            res.update({"out/" + k: out[k].detach().numpy() for k in ("gt_lvis", "pre_lvis", "gt_trace_radiance",
                                                                       "pre_trace_radiance", "sdf_mask")})
            res.update({"trace/" + k: v.detach().numpy() for k, v in trace.items()})
            for nme, p in zip(names, params):
                res["grad_sub/" + nme] = subsample(p.grad)
                res["grad_norm/" + nme] = np.float64(p.grad.double().norm().item())
        opt.step()
        if step in (0, adam_steps - 1):
            for nme, p in zip(names, params):
                res[f"adam{step + 1}_sub/" + nme] = subsample(p)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **res)
    print(name + ".npz written; primary hits", int(res["out/sdf_mask"].sum()), "of", B, "secondary hits",
          int(res["trace/sec_sdf_mask"].sum()), "of", res["trace/sec_sdf_mask"].size, "loss", res["step0/loss"],
          "gt_lvis mean", float(res["out/gt_lvis"][res["out/sdf_mask"]].mean()))


def gen_mateillu_render(fields, renderer, out_dir, name, B, n_samples, n_importance, ray_seed, seeds, adam_steps=3):
    """NeuSRenderer.mateIllu_render (renderer.py:630-726) -> EnvmapMaterialNetwork.forward (inverRender.py:530-598), the
    stage-3 loss of mateIllu.py:152-172, its gradients on the material / illumination network and `adam_steps` optimiser
    steps (mateIllu.py:91-95, 174-176).  The visibility sampler's two uniform draws (inverRender.py:152-153) are recorded."""
    stage2_reference()                                          # the .cuda() shim
    from models import inverRender
    assert inverRender.__file__.startswith(REF + "/")
    sdf, col, var, nerf, ref = build_nets(fields, seeds)
    lvis, indi = fields.Lvis(), fields.IndirectLight()
    lvis(torch.zeros(2, 3), torch.ones(2, 3))
    indi(torch.zeros(2, 3))
    lvis.load_state_dict(to_t(synth.lvis_state_dict(seeds["lvis"])))
    indi.load_state_dict(to_t(synth.indilgt_state_dict(seeds["indilgt"])))
    mat = inverRender.EnvmapMaterialNetwork()
    mat.net_cs(torch.zeros(2, 90))                              # materialise the LazyLinear
    mat.load_state_dict(to_t(synth.mateillu_state_dict(seeds["mateillu"])))
    data = torch.from_numpy(synth.ray_batch(B, seed=ray_seed, n_miss=3))
    rays_o, rays_d, true_rgb, mask_in = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
    a = (rays_d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (rays_o * rays_d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    near, far = mid - 1.0, mid + 1.0
    rnd = renderer.NeuSRenderer(n_samples, n_importance, 0, 4, 1.0, nerf=nerf, sdf_network=sdf, deviation_network=var,
                                color_network=col, refColor_network=ref, lvis_network=lvis, indiLgt_network=indi,
                                mateIllu_network=mat)
    params = list(mat.parameters())
    names = [k for k, _ in mat.named_parameters()]
    opt = torch.optim.Adam(params, lr=5e-4)
    res = {"data": data.numpy(), "B": B, "n_samples": n_samples, "n_importance": n_importance, "ray_seed": ray_seed,
           "lr": 5e-4, "mask_weight": 0.1, **{"seed_" + k: v for k, v in seeds.items()}}
    mask = (mask_in > 0.5).float()                              # mateIllu.py:143-146 with train.mask_weight = 0.1
    real_rand = torch.rand
    glob = inverRender.render_with_sg.__globals__
    real_vis = glob["get_diffuse_visibility"]
    for step in range(adam_steps):
        draws, trace = [], {}

        def rand_hook(*a_, **k_):
            r = real_rand(*a_, **k_)
            draws.append(r.clone())
            return r

        def vis_hook(*a_, **k_):
            v = real_vis(*a_, **k_)
            trace["light_vis"] = v.clone()
            return v

        glob["get_diffuse_visibility"] = vis_hook
        torch.rand = rand_hook
        torch.manual_seed(2000 + step)
        real_cat = record_primary_z(rnd, trace)
        try:
            out = rnd.mateIllu_render(rays_o, rays_d, near, far)
        finally:
            torch.rand = real_rand
            glob["get_diffuse_visibility"] = real_vis
            rnd.cat_z_vals = real_cat
        sdf_mask = out["sdf_mask"]
        # mateIllu.py:152-172
        sdf_mask_sum = mask[sdf_mask].sum() + 1e-5
        rgb_error = (out["rgb"][sdf_mask] - true_rgb[sdf_mask]) * mask[sdf_mask]
        rgb_loss = rgb_error.abs().sum() / sdf_mask_sum
        psnr = 20.0 * torch.log10(1.0 / (((out["rgb"][sdf_mask] - true_rgb[sdf_mask]) ** 2 * mask[sdf_mask]).sum() / (sdf_mask_sum * 3.0)).sqrt())
        loss = rgb_loss + out["encoder_loss"]
        opt.zero_grad()
        loss.backward()
        tag = f"step{step}/"
        assert len(draws) == 2 and draws[0].shape == (128, 32)
        res.update({tag + "u_theta": draws[0].numpy(), tag + "u_phi": draws[1].numpy(), tag + "loss": loss.item(),
                    tag + "rgb_loss": rgb_loss.item(), tag + "encoder_loss": float(out["encoder_loss"]), tag + "psnr": psnr.item()})
        if step == 0:
            for k, v in out.items():
                res["out/" + k] = v.detach().numpy() if torch.is_tensor(v) else np.float64(v)
            res["trace/light_vis"] = trace["light_vis"].numpy()
            res["trace/prim_z"] = trace["prim_z"].numpy()          # (round 3) final depths of the primary rays
            for nme, p in zip(names, params):
                res["grad_sub/" + nme] = subsample(p.grad)
                res["grad_norm/" + nme] = np.float64(p.grad.double().norm().item())
        opt.step()
        if step in (0, adam_steps - 1):
            for nme, p in zip(names, params):
                res[f"adam{step + 1}_sub/" + nme] = subsample(p)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **res)
    print(name + ".npz written; hits", int(res["out/sdf_mask"].sum()), "of", B, "loss", res["step0/loss"], "rgb mean",
          float(res["out/rgb"][res["out/sdf_mask"]].mean()), "lvis_mean", float(res["out/lvis_mean"][res["out/sdf_mask"]].mean()))


def gen_dtu_eval(out_dir, name="dtu_eval_synth"):
    """evaluation/dtu_eval.py eval() of the reference on a synthetic DTU-shaped case (fneus.synth.dtu_eval_scene): the
    Chamfer numbers the product's evaluation/chamfer.py is pinned to.  open3d and trimesh are absent: their FILE I/O
    (read_triangle_mesh / read_point_cloud / write_point_cloud / PointCloud.export) is served from arrays -- no part of the
    evaluated algorithm (sampling, thinning, masking, nearest neighbours: numpy / sklearn) goes through them.  The
    reference shuffles with an unseeded generator (dtu_eval.py:81): it is seeded here, the product is compared with a
    tolerance that covers the shuffle."""
    import tempfile
    from scipy.io import savemat
    scene = synth.dtu_eval_scene(0)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "ObsMask"))
    savemat(os.path.join(tmp, "ObsMask", "ObsMask1_10.mat"), {"ObsMask": scene["ObsMask"], "BB": scene["BB"], "Res": scene["Res"]})
    savemat(os.path.join(tmp, "ObsMask", "Plane1.mat"), {"P": scene["P"].reshape(4, 1)})

    class _Mesh:
        vertices, triangles = scene["vertices"], scene["triangles"]

        def remove_unreferenced_vertices(self):
            return self

    class _Cloud:
        points = scene["stl"]
        colors = None

    o3d = types.ModuleType("open3d")
    o3d.io = types.SimpleNamespace(read_triangle_mesh=lambda p: _Mesh(), read_point_cloud=lambda p: _Cloud(),
                                   write_point_cloud=lambda *a, **k: None)
    o3d.geometry = types.SimpleNamespace(PointCloud=_Cloud)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda x: x)
    tm = types.ModuleType("trimesh")
    tm.PointCloud = lambda pts: types.SimpleNamespace(export=lambda *a, **k: None)
    saved = {k: sys.modules.get(k) for k in ("open3d", "trimesh")}
    sys.modules["open3d"], sys.modules["trimesh"] = o3d, tm
    real_rng = np.random.default_rng
    cwd = os.getcwd()
    try:
        dtu_eval = _load_by_path("ref_dtu_eval", os.path.join(REF, "evaluation", "dtu_eval.py"))
        assert dtu_eval.__file__.startswith(REF + "/")
        sys.modules["ref_dtu_eval"] = dtu_eval          # its multiprocessing pool pickles sample_single_tri by module name
        np.random.default_rng = lambda *a, **k: real_rng(12345)
        os.chdir(tmp)
        dtu_eval.eval("mesh.ply", 1, tmp, tmp)
        mean_d2s, mean_s2d, over_all = [float(x) for x in open(os.path.join(tmp, "result.txt")).read().split()]
    finally:
        os.chdir(cwd)
        np.random.default_rng = real_rng
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), scene_seed=0, mean_d2s=mean_d2s, mean_s2d=mean_s2d, over_all=over_all)
    print(name + ".npz written; d2s", mean_d2s, "s2d", mean_s2d, "overall", over_all)


def gen_raygen(dataset, out_dir, name="raygen_dtu"):
    """Dataset.gen_rays_at / gen_random_rays_at / near_far_from_sphere (dataset.py:115-151, 186-192) called unbound on a
    stub object that carries exactly the attributes they read: a synthetic DTU-like camera set (K^-1, pose), BGR/256
    images and masks.  The random pixel draws are recorded (torch.randint is wrapped), so the fixture pins pixels -> rays."""
    rs = np.random.RandomState(77)
    n_img, H, W = 3, 48, 64
    Ks, poses = [], []
    for i in range(n_img):
        f = 60.0 + 5.0 * i
        K = np.array([[f, 0.3 * i, W / 2 - 0.5 + i], [0, f * 1.02, H / 2 - 0.5 - i], [0, 0, 1]], dtype=np.float64)
        c = rs.standard_normal(3)
        c = 2.8 * c / np.linalg.norm(c)
        zc = -c / np.linalg.norm(c)
        up = np.array([0.1, 1.0, 0.2])
        xc = np.cross(up, zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        pose = np.eye(4)
        pose[:3, :3] = np.stack([xc, yc, zc], 1)          # camera-to-world rotation (columns = camera axes)
        pose[:3, 3] = c
        K4 = np.eye(4)
        K4[:3, :3] = K
        Ks.append(K4)
        poses.append(pose)
    stub = types.SimpleNamespace()
    stub.H, stub.W = H, W
    stub.intrinsics_all_inv = torch.from_numpy(np.linalg.inv(np.stack(Ks)).astype(np.float32))
    stub.pose_all = torch.from_numpy(np.stack(poses).astype(np.float32))
    stub.images = torch.from_numpy((rs.randint(0, 256, size=(n_img, H, W, 3)) / 256.0).astype(np.float32))
    stub.masks = torch.from_numpy((rs.uniform(size=(n_img, H, W, 3)) > 0.4).astype(np.float32))
    res = {"H": H, "W": W, "intrinsics_all_inv": stub.intrinsics_all_inv.numpy(), "pose_all": stub.pose_all.numpy(),
           "images": stub.images.numpy(), "masks": stub.masks.numpy()}
    D = dataset.Dataset
    orig_cuda, orig_randint, orig_avail = torch.Tensor.cuda, torch.randint, torch.cuda.is_available
    draws = []

    def randint_hook(*a, **k):
        r = orig_randint(*a, **k)
        draws.append(r.clone())
        return r

    torch.Tensor.cuda = lambda self, *a, **k: self          # dataset.py:139 hard-codes .cuda()
    torch.randint = randint_hook
    torch.cuda.is_available = lambda: False
    try:
        for lvl in (1, 4):
            ro, rv = D.gen_rays_at(stub, 1, resolution_level=lvl)
            res[f"rays_at_l{lvl}/rays_o"], res[f"rays_at_l{lvl}/rays_v"] = ro.numpy().copy(), rv.numpy().copy()
        torch.manual_seed(123)
        for i, (img, bs) in enumerate(((0, 64), (2, 33))):
            draws.clear()
            out = D.gen_random_rays_at(stub, torch.tensor(img), bs)
            res[f"random_{i}/img_idx"], res[f"random_{i}/pixels_x"], res[f"random_{i}/pixels_y"] = img, draws[0].numpy(), draws[1].numpy()
            res[f"random_{i}/out"] = out.numpy().copy()
            near, far = D.near_far_from_sphere(stub, out[:, :3], out[:, 3:6])
            res[f"random_{i}/near"], res[f"random_{i}/far"] = near.numpy(), far.numpy()
    finally:
        torch.Tensor.cuda, torch.randint, torch.cuda.is_available = orig_cuda, orig_randint, orig_avail
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **res)
    print(name + ".npz written")


def write_shiny_case(case_dir, png, disp, alpha, meta, ball):
    """the files of a Shiny-Blender case from the fixture's arrays (PNG and float TIFF are lossless): used by this generator and,
    with the same arrays, by the tests"""
    from PIL import Image
    os.makedirs(case_dir, exist_ok=True)
    with open(os.path.join(case_dir, "transforms_train.json"), "w") as fp:
        fp.write(meta)
    for i in range(png.shape[0]):
        Image.fromarray(png[i]).save(os.path.join(case_dir, "r_%d.png" % i))
        if ball:
            Image.fromarray(alpha[i]).save(os.path.join(case_dir, "r_%d_alpha.png" % i))
        else:
            Image.fromarray(disp[i]).save(os.path.join(case_dir, "r_%d_disp.tiff" % i))


def gen_raygen_shiny(dataset, out_dir, name="raygen_shiny"):
    """DatasetShiny of the reference itself (models/dataset.py:522-662) on a small Shiny-Blender-format case written here:
    transforms_train.json (camera_angle_x, OpenGL camera-to-world matrices at radius 5.6), r_%d.png colours, r_%d_disp.tiff
    disparities (the `ball` variant: r_%d_alpha.png).  The reference's own rend_util.load_rgb runs (loaded from its file), with
    imageio.imread / tifffile.imread / cv2.imread served from PIL.  Pins: linear colours, masks, intrinsics, poses (centres
    halved, axes flipped), gen_rays_at, gen_random_rays_at (pixel draws recorded), near_far_from_sphere.  The fixture holds the
    file CONTENTS as arrays; the tests write them back to files for the repo's loader."""
    import json, tempfile
    from PIL import Image
    rs = np.random.RandomState(91)
    n_img, H, W = 3, 24, 32
    png = rs.randint(0, 256, size=(n_img, H, W, 3)).astype(np.uint8)
    disp = (rs.uniform(size=(n_img, H, W)) > 0.35).astype(np.float32) * rs.uniform(0.2, 3.0, size=(n_img, H, W)).astype(np.float32)
    alpha = np.repeat((rs.uniform(size=(n_img, H, W, 1)) > 0.4).astype(np.uint8) * 255, 3, axis=-1)
    alpha[:, ::5, ::3] = 100                                    # values below the 0.5 threshold that are not zero
    frames = []
    for i in range(n_img):
        c = rs.standard_normal(3)
        c = 5.6 * c / np.linalg.norm(c)
        z = c / np.linalg.norm(c)                               # OpenGL camera: looks along -z
        x = np.cross(np.array([0.1, 0.2, 1.0]), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, c
        frames.append({"file_path": "r_%d" % i, "transform_matrix": m.tolist()})
    meta = json.dumps({"camera_angle_x": 0.6911112070083618, "frames": frames})
    # readers through PIL, the reference's own rend_util on top of them
    imageio = types.ModuleType("imageio")
    imageio.imread = lambda path, **kw: np.asarray(Image.open(path))
    imageio.plugins = types.SimpleNamespace(freeimage=types.SimpleNamespace(download=lambda: None))
    saved = {k: sys.modules.get(k) for k in ("imageio", "models.rend_util")}
    sys.modules["imageio"] = imageio
    rend_util = _load_by_path("ref_rend_util_for_goldens", os.path.join(REF, "models", "rend_util.py"))
    assert rend_util.__file__.startswith(REF + "/")
    old = (dataset.rend_util, getattr(dataset.tf, "imread", None), getattr(dataset.cv, "imread", None))
    dataset.rend_util = rend_util
    dataset.tf.imread = lambda path: np.array(Image.open(path), dtype=np.float32)
    dataset.cv.imread = lambda path: np.asarray(Image.open(path).convert("RGB"))[..., ::-1].copy()      # BGR like cv2
    orig_cuda, orig_randint, orig_avail = torch.Tensor.cuda, torch.randint, torch.cuda.is_available
    draws = []

    def randint_hook(*a, **k):
        r = orig_randint(*a, **k)
        draws.append(r.clone())
        return r

    class Conf(dict):
        def get_string(self, k):
            return self[k]

    res = {"png": png, "disp": disp, "alpha": alpha, "meta": np.array(meta), "H": H, "W": W}
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.randint = randint_hook
    torch.cuda.is_available = lambda: False
    try:
        with tempfile.TemporaryDirectory() as tmp:
            for ball in (False, True):
                tag = "ball" if ball else "disp"
                case = os.path.join(tmp, "ball_case" if ball else "case")        # ('ball' in data_dir selects the alpha masks)
                write_shiny_case(case, png, disp, alpha, meta, ball)
                ds = dataset.DatasetShiny(Conf(data_dir=case))
                res[f"{tag}/images"], res[f"{tag}/masks"] = ds.images.numpy().copy(), ds.masks.numpy().copy()
                res[f"{tag}/intrinsics_all"], res[f"{tag}/pose_all"] = ds.intrinsics_all.numpy().copy(), ds.pose_all.numpy().copy()
                res[f"{tag}/focal"], res[f"{tag}/n_images"] = float(ds.focal), ds.n_images
                for lvl in (1, 2):
                    ro, rv = ds.gen_rays_at(1, resolution_level=lvl)
                    res[f"{tag}/rays_at_l{lvl}/rays_o"], res[f"{tag}/rays_at_l{lvl}/rays_v"] = ro.numpy().copy(), rv.numpy().copy()
                torch.manual_seed(321)
                draws.clear()
                out = ds.gen_random_rays_at(torch.tensor(2), 48)
                res[f"{tag}/random/pixels_x"], res[f"{tag}/random/pixels_y"] = draws[0].numpy(), draws[1].numpy()
                res[f"{tag}/random/out"] = out.numpy().copy()
                near, far = ds.near_far_from_sphere(out[:, :3], out[:, 3:6])
                res[f"{tag}/random/near"], res[f"{tag}/random/far"] = near.numpy(), far.numpy()
    finally:
        torch.Tensor.cuda, torch.randint, torch.cuda.is_available = orig_cuda, orig_randint, orig_avail
        dataset.rend_util = old[0]
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **res)
    print(name + ".npz written")


FIXTURES = ("units", "render_wmask_b16_n16", "render_wmask_b8_n64", "render_womask_b16_n16_o8", "render_wmask_b16_n16_c0",
            "render_wmask_b256_n32", "render_wmask_b64_n64", "render_wmask_b512_n64", "render_womask_b64_n64_o32",
            "lvis_util_b24_n32", "raygen_dtu", "lvis_render_room_b24_n32",
            "lvis_render_ball_b16_n16", "mateillu_render_b24_n32", "dtu_eval_synth", "lvis_render_room_b128_n64",
            "mateillu_render_b128_n64", "raygen_shiny")


def check_against(old_dir, new_dir, names):
    """every key of a committed fixture must be reproduced bit for bit (new keys may be added)"""
    ok = True
    for name in names:
        old_p, new_p = os.path.join(old_dir, name + ".npz"), os.path.join(new_dir, name + ".npz")
        if not os.path.exists(old_p) or not os.path.exists(new_p):
            continue
        old, new = np.load(old_p), np.load(new_p)
        bad = [k for k in old.files if k not in new.files or not np.array_equal(old[k], new[k])]
        print(f"{name}: {len(old.files)} committed keys, {len(new.files)} regenerated, {len(bad)} differ {bad[:5]}")
        ok = ok and not bad
    return ok


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE, help="directory to write the fixtures to")
    ap.add_argument("--only", default="", help="comma-separated fixture names (default: all)")
    ap.add_argument("--check", action="store_true", help="compare the regenerated fixtures in --out with the committed ones")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    only = set(filter(None, args.only.split(",")))
    want = lambda n: not only or n in only
    torch.set_num_threads(8)
    embedder, fields, renderer, dataset = import_reference(with_dataset=True)
    seeds = {"sdf": 20, "color": 21, "refcolor": 22, "nerf": 23}
    if want("units"):
        gen_units(embedder, fields, renderer, args.out)
    if want("render_wmask_b16_n16"):
        gen_render(fields, renderer, args.out, "render_wmask_b16_n16", B=16, n_samples=16, n_importance=16, n_outside=0,
                   cos_anneal_ratio=1.0, ray_seed=31, n_miss=2, inside_rays=2, mask_weight=0.1, seeds=seeds)
    if want("render_wmask_b8_n64"):
        gen_render(fields, renderer, args.out, "render_wmask_b8_n64", B=8, n_samples=64, n_importance=64, n_outside=0,
                   cos_anneal_ratio=1.0, ray_seed=32, n_miss=1, inside_rays=0, mask_weight=0.1, seeds=seeds)
    if want("render_womask_b16_n16_o8"):
        gen_render(fields, renderer, args.out, "render_womask_b16_n16_o8", B=16, n_samples=16, n_importance=16, n_outside=8,
                   cos_anneal_ratio=0.3, ray_seed=33, n_miss=2, inside_rays=1, mask_weight=0.0, seeds=seeds,
                   white_bkgd=True)
    if want("render_wmask_b16_n16_c0"):
        gen_render(fields, renderer, args.out, "render_wmask_b16_n16_c0", B=16, n_samples=16, n_importance=16, n_outside=0,
                   cos_anneal_ratio=0.0, ray_seed=34, n_miss=0, inside_rays=0, mask_weight=0.1, seeds=seeds,
                   adam_steps=1)
    # ---- round 2 -----------------------------------------------------------------------------------------------
    # BASELINE config 1: 256 rays x (32 + 32) samples, 8 new depths per up-sampling step (per-sample arrays: every 16th ray)
    if want("render_wmask_b256_n32"):
        gen_render(fields, renderer, args.out, "render_wmask_b256_n32", B=256, n_samples=32, n_importance=32, n_outside=0,
                   cos_anneal_ratio=1.0, ray_seed=35, n_miss=12, inside_rays=4, mask_weight=0.1, seeds=seeds, ray_stride=16)
    # the reference configuration's depth (64 + 64) on 64 rays (per-sample arrays: every 4th ray)
    if want("render_wmask_b64_n64"):
        gen_render(fields, renderer, args.out, "render_wmask_b64_n64", B=64, n_samples=64, n_importance=64, n_outside=0,
                   cos_anneal_ratio=1.0, ray_seed=36, n_miss=4, inside_rays=2, mask_weight=0.1, seeds=seeds, ray_stride=4)
    # BASELINE config 2 at full size: 512 rays x (64 + 64) samples, the shape bench.py measures (per-sample arrays: every 32nd ray)
    if want("render_wmask_b512_n64"):
        gen_render(fields, renderer, args.out, "render_wmask_b512_n64", B=512, n_samples=64, n_importance=64, n_outside=0,
                   cos_anneal_ratio=1.0, ray_seed=38, n_miss=24, inside_rays=6, mask_weight=0.1, seeds=seeds, ray_stride=32)
    # womask.conf's sample counts (64 + 64 + 32 outside) on 64 rays (per-sample arrays: every 4th ray)
    if want("render_womask_b64_n64_o32"):
        gen_render(fields, renderer, args.out, "render_womask_b64_n64_o32", B=64, n_samples=64, n_importance=64, n_outside=32,
                   cos_anneal_ratio=0.7, ray_seed=39, n_miss=4, inside_rays=2, mask_weight=0.0, seeds=seeds, white_bkgd=False,
                   ray_stride=4)
    if want("lvis_util_b24_n32"):
        gen_lvis_util(fields, renderer, args.out, "lvis_util_b24_n32", B=24, n_samples=32, n_importance=32, ray_seed=37, seeds=seeds)
    if want("raygen_dtu"):
        gen_raygen(dataset, args.out)
    if want("raygen_shiny"):
        gen_raygen_shiny(dataset, args.out)
    # ---- stage 2 (config 3): lvis_render + cal_indiLgt, loss, gradients and Adam steps of Lvis + IndirectLight -------------
    seeds2 = dict(seeds, lvis=24, indilgt=25)
    if want("lvis_render_room_b24_n32"):        # a room of radius 0.55 seen from within: chords beyond 1 leave without a hit
        gen_lvis_render(fields, renderer, args.out, "lvis_render_room_b24_n32", B=24, n_samples=32, n_importance=32,
                        ray_seed=41, seeds=seeds2, room=True, bias=0.7, warp=(3, 0.1), outside_rays=3)
    if want("lvis_render_ball_b16_n16"):        # the convex ball: secondary rays leave without a hit, visibility in (0.5, 1]
        gen_lvis_render(fields, renderer, args.out, "lvis_render_ball_b16_n16", B=16, n_samples=16, n_importance=16,
                        ray_seed=42, seeds=seeds2, room=False, adam_steps=1)
    if want("lvis_render_room_b128_n64"):       # the stage's configured depth (64 + 64) on 128 primary rays / ~460 secondary rays
        gen_lvis_render(fields, renderer, args.out, "lvis_render_room_b128_n64", B=128, n_samples=64, n_importance=64,
                        ray_seed=44, seeds=seeds2, room=True, bias=0.7, warp=(3, 0.1), outside_rays=12, adam_steps=1)
    # ---- stage 3 (configs 4/5): mateIllu_render + EnvmapMaterialNetwork, loss, gradients, Adam steps -----------------------
    if want("mateillu_render_b24_n32"):
        gen_mateillu_render(fields, renderer, args.out, "mateillu_render_b24_n32", B=24, n_samples=32, n_importance=32,
                            ray_seed=43, seeds=dict(seeds2, mateillu=26))
    if want("mateillu_render_b128_n64"):
        gen_mateillu_render(fields, renderer, args.out, "mateillu_render_b128_n64", B=128, n_samples=64, n_importance=64,
                            ray_seed=45, seeds=dict(seeds2, mateillu=26), adam_steps=1)
    if want("dtu_eval_synth"):
        gen_dtu_eval(args.out)
    if args.check:
        ok = check_against(HERE, args.out, FIXTURES)
        print("committed fixtures reproduced bit for bit" if ok else "MISMATCH against the committed fixtures")
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
