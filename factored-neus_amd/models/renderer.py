"""NeuSRenderer with the reference's API (models/renderer.py:80-500), stage-1 hot path on HIP kernels.

render() produces the reference's 13-key dict.  Every per-ray-sample operation runs in libfneus_hip.so:
  sampler  : fneus_sdf_fwd (K1) + fneus_upsample / fneus_merge (K6)              renderer.py:425-449
  core     : fneus_sections, fneus_sdf_fwd_grad (K2), fneus_color_fwd (K4), fneus_composite_fwd (K5)   :208-389
  backward : the hand-written adjoints, reached through torch.autograd (fneus/autograd.py)
torch only allocates, wires the graph, runs the <= 2-samples-per-ray RefColor branch and the scalar losses.

Differences from the reference that a caller can observe:
  * tensors live on the device of the networks (no global default-device trick, exp_runner.py:638-641);
  * the surface branch is evaluated at fixed shape (all rays, masked) so there is no host sync per step; rows outside
    sdf_mask are 1.0 exactly as in the reference (renderer.py:280-282).
"""
from __future__ import annotations

import os

import numpy as np

import torch

from fneus import ops
from models.mesh import extract_fields, extract_geometry      # noqa: F401  module-level API of renderer.py:14-40
from fneus.autograd import CompositeFn, OutsideAlphaFn, OutsideAlphaSelFn, RaySamples, SurfaceGatherFn, Stage1LossFn


def sample_pdf(bins, weights, n_samples, det=False):
    """reference module-level helper (renderer.py:43-77); the hot path uses the fused fneus_upsample instead."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=bins.device)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples], device=bins.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)


class _LazySVal:
    """`s_val` = 1 / inv_s for every sample (renderer.py:380) is a logging quantity: it is materialised on demand instead
    of costing three element-wise launches in every training step."""

    def __init__(self, deviation_network, B, n):
        self.net, self.B, self.n = deviation_network, B, n

    def per_sample(self):
        with torch.no_grad():
            return (1.0 / self.net.inv_s()).expand(self.B * self.n, 1)

    def per_ray(self):      # mean over the samples of a ray (renderer.py:488) of identical values
        with torch.no_grad():
            return (1.0 / self.net.inv_s()).expand(self.B, 1)


SAMPLER_FUSED = os.environ.get("FNEUS_SAMPLER_FUSED", "1") != "0"

class NeuSRenderer:
    def __init__(self, n_samples, n_importance, n_outside, up_sample_steps, perturb, nerf=None, sdf_network=None,
                 deviation_network=None, color_network=None, refColor_network=None, lvis_network=None,
                 indiLgt_network=None, mateIllu_network=None):
        self.nerf = nerf
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.refColor_network = refColor_network
        self.lvis_network = lvis_network
        self.indiLgt_network = indiLgt_network
        self.mateIllu_network = mateIllu_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb

    # ---- reference-compatible pieces (renderer.py:152-205) ------------------------------------------------------
    def up_sample(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s):
        B, m = z_vals.shape
        return ops.upsample(rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous(),
                            sdf.reshape(B, m).contiguous(), n_importance, float(inv_s))

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, last=False):
        B, k = new_z_vals.shape
        if last:
            z, _ = ops.merge(z_vals.contiguous(), None, new_z_vals.contiguous(), None)
            return z, sdf
        new_sdf = self.sdf_network.sdf_samples(
            RaySamples(rays_o, rays_d, new_z_vals.reshape(-1).contiguous(), k)).reshape(B, k)
        return ops.merge(z_vals.contiguous(), sdf.contiguous(), new_z_vals.contiguous(), new_sdf)

    def _hierarchical_z(self, rays_o, rays_d, z_vals):
        """renderer.py:433-446.  Default: the merge of step i and the up_sample of step i + 1 (and the last step's merge) are
        ONE launch (fneus_merge_upsample; bit-identical to the separate calls, 7 launches instead of 11 per render);
        FNEUS_SAMPLER_FUSED=0, a single step, or an instance whose up_sample / cat_z_vals were replaced (tests hook them):
        the reference's call sequence through the two methods."""
        B, n = z_vals.shape
        steps = self.up_sample_steps
        k = self.n_importance // steps
        own = "up_sample" not in self.__dict__ and "cat_z_vals" not in self.__dict__
        with torch.no_grad():
            sdf = self.sdf_network.sdf_samples(RaySamples(rays_o, rays_d, z_vals.reshape(-1).contiguous(), n)).reshape(B, n)
            if not (own and steps >= 2 and SAMPLER_FUSED and n + k * steps <= 256):
                for i in range(steps):
                    new_z = self.up_sample(rays_o, rays_d, z_vals, sdf, k, 64 * 2 ** i)
                    z_vals, sdf = self.cat_z_vals(rays_o, rays_d, z_vals, new_z, sdf, last=(i + 1 == steps))
                return z_vals
            ro, rd = rays_o.contiguous(), rays_d.contiguous()
            z_vals, sdf = z_vals.contiguous(), sdf.contiguous()
            new_z = ops.upsample(ro, rd, z_vals, sdf, k, 64.0)
            # round 6: ALL remaining steps in one launch -- a workgroup keeps its two rays through the steps (None: not this shape)
            fused = self.sdf_network.sdf_merge_upsample_steps(ro, rd, z_vals, sdf, new_z.contiguous(), [float(64 * 2 ** i) for i in range(1, steps)],
                                                              k, 2.0 / self.n_samples) if steps >= 3 else None
            if fused is not None:
                z_final, dists, mid_z = fused
                self._final_sections = (z_final, 2.0 / self.n_samples, dists, mid_z)
                return z_final
            for i in range(1, steps):
                # round 6: the SDF evaluation of the step's new depths and the merge + next up_sample in ONE launch (a tile of the
                # evaluation is whole rays: the workgroup that evaluated it merges them); None: a shape that launch does not take
                fused = self.sdf_network.sdf_merge_upsample(ro, rd, z_vals, sdf, new_z.contiguous(), float(64 * 2 ** i), k,
                                                            last=(i + 1 == steps), sample_dist=2.0 / self.n_samples)
                if fused is not None:
                    z_vals, sdf, new_z, z_final, dists, mid_z = fused
                    continue
                new_sdf = self.sdf_network.sdf_samples(RaySamples(rays_o, rays_d, new_z.reshape(-1), k)).reshape(B, k)
                # (the last launch also writes the sections of the final depths: render_core asks for them next)
                z_vals, sdf, new_z, z_final, dists, mid_z = ops.merge_upsample(ro, rd, z_vals, sdf, new_z, new_sdf.contiguous(),
                                                                               float(64 * 2 ** i), k, last=(i + 1 == steps),
                                                                               sample_dist=2.0 / self.n_samples)
        self._final_sections = (z_final, 2.0 / self.n_samples, dists, mid_z)
        return z_final

    # ---- render_core_outside (renderer.py:112-149): inverted-sphere background NeRF++, womask configs only --------
    def render_core_outside(self, rays_o, rays_d, z_vals, sample_dist, nerf, background_rgb=None, full=True, z_core=None):
        """The background network runs on the fused K7 kernels (models/fields.py NeRF -> fneus_nerf_bg_fwd / _bwd); the
        inverted-sphere points (fneus_outside_points) and softplus / sigmoid / alpha (fneus_outside_alpha_fwd / _bwd) are one
        launch each; the blend with the foreground is inside the HIP compositing kernels.  full=False (what render() asks
        for): only `sampled_color` and `alpha`, the two entries the reference's render() consumes (renderer.py:455-458) --
        this branch's own weights and colour are dead code on that path."""
        B, n = z_vals.shape
        if z_core is not None and not full and n <= 256 and ops.bg_select_enabled():
            # render() hands over the depths render_core works on: the network runs only where render_core uses its value
            # (ops.outside_select; samples inside the unit sphere take the foreground alone, renderer.py:350-356)
            s = ops.outside_select(rays_o.contiguous(), rays_d.contiguous(), z_core.contiguous(), z_vals.contiguous(), sample_dist,
                                   count=nerf.select_count(B * n) if hasattr(nerf, "select_count") else None)
            density, rgb_raw = nerf(s.pts4, s.dirs, n_active=s.count)
            alpha, rgb = OutsideAlphaSelFn.apply(density.reshape(-1), rgb_raw, s)
            return {"sampled_color": rgb, "alpha": alpha}
        pts4, dirs, dists = ops.outside_points(rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous(), sample_dist)
        density, rgb_raw = nerf(pts4, dirs)
        alpha, rgb = OutsideAlphaFn.apply(density.reshape(-1), rgb_raw, dists.reshape(-1))
        alpha, rgb = alpha.reshape(B, n), rgb.reshape(B, n, 3)
        if not full:
            return {"sampled_color": rgb, "alpha": alpha}
        one = torch.ones([B, 1], device=z_vals.device)
        weights = alpha * torch.cumprod(torch.cat([one, 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
        color = (weights[:, :, None] * rgb).sum(dim=1)
        if background_rgb is not None:
            color = color + background_rgb * (1.0 - weights.sum(dim=-1, keepdim=True))
        return {"color": color, "sampled_color": rgb, "alpha": alpha, "weights": weights}

    def _ones_b3(self, B, device):
        key = (B, str(device))
        cache = self.__dict__.setdefault("_ones_cache", {})
        if key not in cache:
            cache[key] = torch.ones(B, 3, device=device)
        return cache[key]

    def _sections(self, z_vals, sample_dist):
        """section lengths and mid points (renderer.py:223-226): the sampler's last launch has written them for ITS result"""
        fs = self.__dict__.pop("_final_sections", None)
        if fs is not None and fs[0] is z_vals and fs[1] == sample_dist:
            return fs[2], fs[3]
        return ops.sections(z_vals.contiguous(), sample_dist)

    # ---- render_core (renderer.py:208-389) ------------------------------------------------------------------------
    def render_core(self, rays_o, rays_d, z_vals, sample_dist, sdf_network, deviation_network, color_network,
                    refColor_network, background_alpha=None, background_sampled_color=None, background_rgb=None,
                    cos_anneal_ratio=0.0, loss_args=None, background_fn=None):
        """background_fn() -> (alpha, sampled_color): the background evaluated HERE, right in front of the compositing, instead of
        ahead of the call (render() passes it: with the background network's forward behind the SDF network's, autograd runs its
        backward first and the SDF network's weight-gradient launch can take its products along)"""
        B, n = z_vals.shape
        train = torch.is_grad_enabled()
        dists, mid_z = self._sections(z_vals, sample_dist)
        samples = RaySamples(rays_o, rays_d, mid_z.reshape(-1), n)
        # (feat goes to the colour network and the surface gather only: both read the SDF stash's feature planes, so a chip-filling
        #  training launch writes no fp32 feature rows -- round 6)
        sdf, feat, normal = sdf_network.value_feature_normal(samples, train, feat_rows=False)
        rgb = color_network.color_samples(samples, normal, feat, sdf_network, train)
        if background_fn is not None:
            background_alpha, background_sampled_color = background_fn()
        # inv_s = clip(exp(10 variance)) (renderer.py:245) is applied inside the compositing kernels
        (color, weights, wsum, wpair, eik_num, wmax, cdf, inside, eik_den, min_idx, sdf_mask_u8) = CompositeFn.apply(
            sdf, normal, rgb, deviation_network.variance, rays_o, rays_d, mid_z, dists,
            cos_anneal_ratio if torch.is_tensor(cos_anneal_ratio) else float(cos_anneal_ratio),
            background_alpha, background_sampled_color,
            # renderer.py:367-368 `color + background_rgb * (1 - weights_sum)` for a constant background: inside the kernels
            background_rgb if (background_rgb is not None and not background_rgb.requires_grad) else None)
        sdf_mask = sdf_mask_u8.view(torch.bool)          # the kernel writes exactly 0 / 1: reinterpret, no cast kernel
        if background_rgb is not None and background_rgb.requires_grad:
            color = color + background_rgb * (1.0 - wsum[:, None])

        # surface branch at fixed shape (renderer.py:284-343): the two samples bracketing the first sign change
        # constant, never written on the RefColor paths (they produce fresh tensors): allocated once per batch size
        ones = self._ones_b3(B, z_vals.device) if refColor_network is not None else torch.ones(B, 3, device=z_vals.device)
        specular_color = diffuse_color = surface_color = ones
        losses = None
        if refColor_network is not None:
            feat_sel, normal_sel, t_sel, _sel = SurfaceGatherFn.apply(feat, normal, mid_z, min_idx, sdf_mask_u8,
                                                                      sdf_network._ws, refColor_network.warm_ranges(False))
            # 2 samples per ray in the (rays_o, rays_d, t) form of the fused kernels (pts = o + d * t, renderer.py:322)
            surf = RaySamples(rays_o, rays_d, t_sel, 2)
            diffuse, spec = refColor_network.heads(surf, feat_sel, normal_sel, train)
            if loss_args is not None:     # shading, blend, losses and their gradients in one launch (training step)
                true_rgb, mask_in, igr_w, mask_w, surf_w = loss_args[:5]
                reduce_norms = loss_args[5] if len(loss_args) > 5 else None
                loss, lvec, surface_color, specular_color, diffuse_color = Stage1LossFn.apply(
                    color, wsum, eik_num, wpair, diffuse, spec, eik_den, true_rgb, mask_in, sdf_mask_u8, float(igr_w),
                    float(mask_w), float(surf_w), reduce_norms, refColor_network.warm_ranges(True))
                losses = {"loss": loss, "color_loss": lvec[1], "surface_loss": lvec[2], "eikonal_loss": lvec[3],
                          "mask_loss": lvec[4], "psnr": lvec[5]}
            else:
                ref = refColor_network.shade(diffuse, spec)
                w_lo, w_hi = wpair[:, 0:1] + 1e-5, wpair[:, 1:2] + 1e-5

                def blend(v):
                    v = v.reshape(B, 2, 3)
                    return torch.where(sdf_mask[:, None], (v[:, 0] * w_lo + v[:, 1] * w_hi) / (w_lo + w_hi), ones)

                specular_color, diffuse_color, surface_color = blend(ref["specular_rgb"]), blend(ref["diffuse_rgb"]), blend(ref["rgb"])
        elif loss_args is not None:
            raise NotImplementedError("the fused training loss needs the RefColor head (surface term)")
        gradient_error = losses["eikonal_loss"] if losses is not None else eik_num.sum() / (eik_den.sum() + 1e-5)   # renderer.py:370-372

        return {
            "color": color, "surface_color": surface_color, "sdf_mask": sdf_mask, "sdf": sdf[:, None], "dists": dists,
            "gradients": normal.reshape(B, n, 3), "s_val": _LazySVal(deviation_network, B, n), "mid_z_vals": mid_z,
            "weights": weights, "cdf": cdf, "gradient_error": gradient_error, "inside_sphere": inside,
            "specular_color": specular_color, "diffuse_color": diffuse_color, "weight_max": wmax, "weight_sum": wsum,
            "losses": losses,
        }

    # ---- render (renderer.py:391-500) -------------------------------------------------------------------------------
    def render(self, rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0,
               z_vals_override=None, loss_args=None):
        """near / far [B,1], or both None for the unit-sphere bounds of dataset.py:186-192 (computed on the device).
        loss_args = (true_rgb [B,3], mask [B,1], igr_weight, mask_weight, surface_weight[, reduce_norms]): also evaluate
        the training losses of exp_runner.py:141-177 (fused with the surface shading, fneus_stage1_loss) ->
        out["losses"].  reduce_norms(norms[4]) -> norms[4]: data parallel, sums the loss normalisers over the ranks;
        the loss terms are then this rank's share of the global batch's (sum them, and the gradients, over the ranks)."""
        dev = rays_o.device
        rays_o, rays_d = rays_o.float().contiguous(), rays_d.float().contiguous()
        B = len(rays_o)
        sample_dist = 2.0 / self.n_samples
        z_vals_outside = None
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        t_rand = torch.rand([B, 1], device=dev) if perturb > 0 else None
        # z = near + (far - near) * linspace(0, 1, n) (+ jitter), renderer.py:393-409; near = None: unit-sphere bounds
        # (dataset.py:186-192) computed in the same launch
        if near is None:
            z_vals = ops.ray_setup(rays_o, rays_d, self.n_samples, t_rand=t_rand)
        else:
            z_vals = ops.ray_setup(rays_o, rays_d, self.n_samples, near=near.float().reshape(-1).contiguous(),
                                   far=far.float().reshape(-1).contiguous(), t_rand=t_rand)
        if self.n_outside > 0:
            # renderer.py:397-400, 411-419 in one launch (fneus_outside_z): linspace(1e-3, 1 - 1/(n+1), n), jittered inside its cells,
            # flipped, far / t + 1 / n_samples (far = None: the unit-sphere bound of the rays, computed there)
            u_out = torch.rand([B, self.n_outside], device=dev) if perturb > 0 else None
            z_vals_outside = ops.outside_z(rays_o, rays_d, self.n_outside, self.n_samples,
                                           far=None if far is None else far.float().reshape(-1).contiguous(), u=u_out)
        # networks changed since the last call (optimiser step): fold weight-norm and re-pack once -- all of them in one
        # fneus_refresh_multi call (ops.batched_refresh; with FNEUS_OVERLAP bit 1 the sampler's network first, the others beside it)
        if ops.OVERLAP_MASK & 1:
            self.sdf_network.refresh()
            with ops.on_side_stream(1):
                self.color_network.refresh()
                if self.refColor_network is not None:
                    self.refColor_network.refresh()
                if self.n_outside > 0 and self.nerf is not None and hasattr(self.nerf, "refresh"):
                    self.nerf.refresh()
        else:
            with ops.batched_refresh():
                self.sdf_network.refresh()
                self.color_network.refresh()
                if self.refColor_network is not None:
                    self.refColor_network.refresh()
                if self.n_outside > 0 and self.nerf is not None and hasattr(self.nerf, "refresh"):
                    self.nerf.refresh()
        n = self.n_samples
        if self.n_importance > 0:
            if z_vals_override is not None:
                z_vals = z_vals_override
            else:
                z_vals = self._hierarchical_z(rays_o, rays_d, z_vals.contiguous())
            n = self.n_samples + self.n_importance
        ops.overlap_join()                   # the packs issued beside the sampler are needed from here on
        background_fn = None
        if self.n_outside > 0:                                                    # renderer.py:452-458
            def background_fn():
                # sort(cat(z_vals, z_vals_outside)) of two sorted rows = one stable rank merge (fneus_merge)
                z_vals_feed, _ = ops.merge(z_vals.contiguous(), None, z_vals_outside.expand(B, -1).contiguous(), None)
                ret_outside = self.render_core_outside(rays_o, rays_d, z_vals_feed, sample_dist, self.nerf, full=False, z_core=z_vals)
                return ret_outside["alpha"], ret_outside["sampled_color"]
        ret = self.render_core(rays_o, rays_d, z_vals, sample_dist, self.sdf_network, self.deviation_network,
                               self.color_network, self.refColor_network, background_rgb=background_rgb,
                               background_fn=background_fn, cos_anneal_ratio=cos_anneal_ratio, loss_args=loss_args)
        weights = ret["weights"]
        return {
            "color_fine": ret["color"],
            "surface_color": ret["surface_color"],
            "sdf_mask": ret["sdf_mask"],
            "s_val": None if loss_args is not None else ret["s_val"].per_ray(),    # logging quantity: skipped in training steps
            "cdf_fine": ret["cdf"],
            "weight_sum": ret["weight_sum"][:, None],
            "weight_max": ret["weight_max"][:, None],
            "gradients": ret["gradients"],
            "weights": weights,
            "gradient_error": ret["gradient_error"],
            "inside_sphere": ret["inside_sphere"],
            "specular_color": ret["specular_color"],
            "diffuse_color": ret["diffuse_color"],
            "losses": ret["losses"],
            # extras (not in the reference dict; used by the parity tests)
            "_z_vals": z_vals, "_sdf": ret["sdf"], "_mid_z_vals": ret["mid_z_vals"],
        }

    def lvis_mateIllu_render_util(self, rays_o, rays_d, near, far, z_vals_override=None, need_inside=True):
        """renderer.py:503-564, the entry of the stage-2 / stage-3 renderers (lvis_render, mateIllu_render): unperturbed
        hierarchical sampling, SDF at the section mid-points and the per-ray inside-sphere mask.  Geometry is frozen in
        those stages (lvis.py:78-92 optimises the visibility / indirect-light networks only), so the SDF comes from the
        no-grad K1 kernel: fneus_ray_setup + 4 x (fneus_upsample, fneus_sdf_fwd, fneus_merge) + fneus_sections + fneus_sdf_fwd."""
        B = rays_o.shape[0]
        sample_dist = 2.0 / self.n_samples
        rays_o, rays_d = rays_o.detach().float().contiguous(), rays_d.detach().float().contiguous()
        self.sdf_network.refresh()
        with torch.no_grad():
            cf = lambda t: None if t is None else t.detach().float().reshape(B, 1).contiguous()
            z_vals = ops.ray_setup(rays_o, rays_d, self.n_samples, near=cf(near), far=cf(far))
            n = self.n_samples
            if self.n_importance > 0:
                # z_vals_override (tests): the reference's own final depths instead of this sampler's ("teacher forcing", as
                # render() offers it: the inverse CDF is ill conditioned at flat stretches and everything behind the first
                # zero crossing -- hit point, PE10 of it, ReLU networks -- amplifies a moved depth)
                z_vals = self._hierarchical_z(rays_o, rays_d, z_vals.contiguous()) if z_vals_override is None \
                    else z_vals_override.detach().float().contiguous()
                n = self.n_samples + self.n_importance
            dists, mid_z = self._sections(z_vals, sample_dist)
            sdf = self.sdf_network.sdf_samples(RaySamples(rays_o, rays_d, mid_z.reshape(-1), n))
            inside_any = None
            if need_inside:     # (fneus_ray_hit tests the same samples itself when it is not handed the mask: the fixed-shape
                pts = rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]          # steps skip these six launches)
                inside_any = (torch.linalg.norm(pts, ord=2, dim=-1) < 1.0).any(dim=-1)
        return {"n_samples": n, "mid_z_vals": mid_z, "sdf": sdf[:, None], "inside_sphere_mask": inside_any}

    def lvis_render(self, rays_o, rays_d, near, far, u_theta=None, u_z=None, trace=None, fixed_shape=False, z_vals_override=None,
                    raw=False):
        """renderer.py:567-627: visibility / traced radiance of 4 secondary rays per visible surface point, and the
        predictions of the Lvis / IndirectLight networks.  Rows of rays without a surface hit hold 1.
        fneus_ray_hit finds the hit points.  Default: they are compacted (one host read of the hit count per step, as the
        reference's `if n_sdf_mask > 0`) and handed to cal_indiLgt.  fixed_shape=True: every ray is treated as a hit point
        (rays without one start their secondary rays at the ray origin) and masked afterwards -- a few per cent more work,
        no host synchronisation and no data-dependent shape: the step can be captured in a hipGraph (fneus/trainer2.py).
        u_theta, u_z [hits, 4] (or [B, 4] at fixed shape): the uniform draws (tests)."""
        from models.calLvis import cal_indiLgt
        B = len(rays_o)
        dev = rays_o.device
        M = 4
        util = self.lvis_mateIllu_render_util(rays_o, rays_d, near, far, z_vals_override=z_vals_override, need_inside=not fixed_shape)
        n = util["n_samples"]
        rays_o, rays_d = rays_o.detach().float().contiguous(), rays_d.detach().float().contiguous()
        with torch.no_grad():
            hit = ops.ray_hit(rays_o, rays_d, util["mid_z_vals"], util["sdf"].reshape(B, n).contiguous(),
                              inside_mask=util["inside_sphere_mask"])
            sdf_mask = hit["sdf_mask"].view(torch.bool)
        if fixed_shape:
            with torch.no_grad():
                pts_surf = hit["pts_surf"]
                _, _, n_surf = self.sdf_network.value_feature_normal(RaySamples(pts=pts_surf), False)
            res = cal_indiLgt(pts_surf, n_surf, self.sdf_network, self.deviation_network, self.color_network,
                              self.lvis_network, self.indiLgt_network, u_theta=u_theta, u_z=u_z, trace=trace, point_mask=sdf_mask)
            if raw:      # (the caller applies the mask itself -- fneus_stage2_loss: rows of rays without a hit are placeholders)
                return dict(res, sdf_mask=sdf_mask)
            one = torch.ones((), device=dev)
            out = {"sdf_mask": sdf_mask}
            for k in ("gt_lvis", "pre_lvis"):
                out[k] = torch.where(sdf_mask[:, None], res[k], one)
            for k in ("gt_trace_radiance", "pre_trace_radiance"):
                out[k] = torch.where(sdf_mask[:, None, None], res[k], one)
            return out
        with torch.no_grad():
            idx = sdf_mask.nonzero(as_tuple=True)[0]
        out = {"gt_lvis": torch.ones(B, M, device=dev), "pre_lvis": torch.ones(B, M, device=dev),
               "gt_trace_radiance": torch.ones(B, M, 3, device=dev), "pre_trace_radiance": torch.ones(B, M, 3, device=dev),
               "sdf_mask": sdf_mask}
        if idx.numel() > 0:
            with torch.no_grad():
                pts_surf = hit["pts_surf"][idx].contiguous()
                _, _, n_surf = self.sdf_network.value_feature_normal(RaySamples(pts=pts_surf), False)
            res = cal_indiLgt(pts_surf, n_surf, self.sdf_network, self.deviation_network, self.color_network,
                              self.lvis_network, self.indiLgt_network, u_theta=u_theta, u_z=u_z, trace=trace)
            if trace is not None:
                trace.update(pts_surf=pts_surf, normal=n_surf)
            for k in ("gt_lvis", "pre_lvis", "gt_trace_radiance", "pre_trace_radiance"):
                out[k] = out[k].index_copy(0, idx, res[k].to(out[k].dtype))
        return out

    def mateIllu_render(self, rays_o, rays_d, near, far, u_theta=None, u_phi=None, fixed_shape=False, z_vals_override=None,
                        keys=None, raw=False):
        """renderer.py:630-726: stage 3.  Geometry (SDF), the RefColor head, Lvis and IndirectLight are frozen inputs
        (mateIllu.py:83-95 trains the EnvmapMaterialNetwork only): hit points by fneus_ray_hit, normal + feature by K2, the
        diffuse / specular split by the fused RefColor heads, all without stash.  Rows of rays without a hit hold 1.
        fixed_shape=True: every ray is evaluated (rays without a hit at their origin) and masked afterwards, the latent
        sparsity term averages over the hit points only: same values and gradients, no host synchronisation, capturable
        in a hipGraph (fneus/trainer3.py).
        u_theta, u_phi [128, 32]: the uniform draws of the visibility sampler (inverRender.py:152-153; tests).
        keys: the per-ray entries the caller reads (None = the reference's whole dict).  The training step reads `rgb` (plus
        `sdf_mask` and the loss terms, which are always there): every other entry is tone mapping, a fill and a select on [B, 3]
        tensors, a launch each -- 35 launches of the fixed-shape step.
        raw (with fixed_shape): the rows of rays without a hit are left as the networks returned them (placeholders) instead of 1 --
        for a caller that masks with `sdf_mask` itself (fneus_stage3_loss does): a fill and a select per entry, and the select's
        backward, are not launched."""
        from models.inverRender import srgb_to_linear
        B = len(rays_o)
        dev = rays_o.device
        util = self.lvis_mateIllu_render_util(rays_o, rays_d, near, far, z_vals_override=z_vals_override, need_inside=not fixed_shape)
        n = util["n_samples"]
        rays_o, rays_d = rays_o.detach().float().contiguous(), rays_d.detach().float().contiguous()
        with torch.no_grad():
            hit = ops.ray_hit(rays_o, rays_d, util["mid_z_vals"], util["sdf"].reshape(B, n).contiguous(),
                              inside_mask=util["inside_sphere_mask"])
            sdf_mask = hit["sdf_mask"].view(torch.bool)
            idx = None if fixed_shape else sdf_mask.nonzero(as_tuple=True)[0]
        ray_keys = ("rgb", "env_rgb", "indir_rgb", "diffuse_albedo", "specular_albedo", "diffuse_rgb", "specular_rgb", "roughness",
                    "lvis_mean")
        want = None if keys is None else set(keys)
        need = lambda k: want is None or k in want
        extra_keys = ("gt_specular_linear", "gt_diffuse_srgb", "n_out")
        ray_keys = tuple(k for k in ray_keys if need(k))
        one3 = lambda: torch.ones(B, 3, device=dev)
        raw = bool(raw and fixed_shape)
        out = {} if raw else {k: one3() for k in ray_keys + tuple(k for k in extra_keys if need(k))}
        if need("roughness") and not raw:
            out["roughness"] = torch.ones(B, 1, device=dev)
        out.update(sdf_mask=sdf_mask, diffuse_loss=0, specular_loss=0, encoder_loss=0, smooth_loss=0)
        if fixed_shape or idx.numel() > 0:
            with torch.no_grad():
                pts_surf = hit["pts_surf"] if fixed_shape else hit["pts_surf"][idx].contiguous()
                rays_surf = rays_d if fixed_shape else rays_d[idx].contiguous()
                surf = RaySamples(pts=pts_surf, dirs=rays_surf)
                _, f_surf, n_surf = self.sdf_network.value_feature_normal(surf, False)
                self.refColor_network.refresh()
                ref, specular_linear = None, None
                if need("gt_specular_linear") or need("gt_diffuse_srgb"):     # (logged only: the networks below do not read them)
                    diffuse, spec = self.refColor_network.heads(surf, f_surf, n_surf, False)
                    ref = self.refColor_network.shade(diffuse, spec)
                    specular_linear = srgb_to_linear(ref["specular_rgb"])
                # (a frozen MLP on the points the material network encodes anyway: EnvmapMaterialNetwork.forward takes its layers
                # into the launches of its own two MLPs when it can)
                if getattr(self.mateIllu_network, "resolves_deferred_indirect", False) and hasattr(self.indiLgt_network, "deferred") \
                        and pts_surf.is_cuda and not any(p.requires_grad for p in self.indiLgt_network.parameters()):
                    indiLgt = self.indiLgt_network.deferred(pts_surf)
                else:
                    indiLgt = self.indiLgt_network(pts_surf)
            m = self.mateIllu_network(pts_surf, rays_surf, n_surf, f_surf, specular_linear, indiLgt, self.lvis_network,
                                      u_theta=u_theta, u_phi=u_phi, point_mask=sdf_mask if fixed_shape else None, want=want)
            extra = tuple((k, v) for k, v in (("gt_specular_linear", specular_linear),
                                              ("gt_diffuse_srgb", None if ref is None else ref["diffuse_rgb"]), ("n_out", n_surf)) if need(k))
            if raw:
                out.update({k: m[k] for k in ray_keys})
                out.update(dict(extra))
            elif fixed_shape:
                sel = sdf_mask[:, None]
                for k in ray_keys:
                    out[k] = torch.where(sel, m[k], out[k])
                for k, v in extra:
                    out[k] = torch.where(sel, v, out[k])
            else:
                for k in ray_keys:
                    out[k] = out[k].index_copy(0, idx, m[k].to(out[k].dtype))
                for k, v in extra:
                    out[k] = out[k].index_copy(0, idx, v)
            for k in ("diffuse_loss", "specular_loss", "encoder_loss", "smooth_loss"):
                out[k] = m[k]
        return out

    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.0):
        """renderer.py:729-734: iso-surface of -sdf at `threshold`; the grid goes through K1 (fneus_sdf_fwd), the surface
        is extracted on the device by models/mesh.py (marching tetrahedra; PyMCubes is not a dependency).
        -> (vertices [V,3] float64 numpy, triangles [T,3] int64 numpy)"""
        from models.mesh import marching_tetrahedra
        u = self.extract_sdf_grid(bound_min, bound_max, resolution)
        verts, tris = marching_tetrahedra(u, threshold)
        b_min = np.asarray([float(bound_min[i]) for i in range(3)])
        b_max = np.asarray([float(bound_max[i]) for i in range(3)])
        vertices = verts.cpu().numpy().astype(np.float64) / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :]
        return vertices, tris.cpu().numpy()

    def extract_sdf_grid(self, bound_min, bound_max, resolution):
        """-SDF on a regular grid (extract_fields with the query function of renderer.py:733), chunked through K1."""
        dev = self.sdf_network.lin0.bias.device
        self.sdf_network.refresh()
        xs = [torch.linspace(float(bound_min[i]), float(bound_max[i]), resolution, device=dev) for i in range(3)]
        u = torch.empty(resolution, resolution, resolution, device=dev)
        N = 64
        with torch.no_grad():
            for xi in range(0, resolution, N):
                for yi in range(0, resolution, N):
                    xx, yy, zz = torch.meshgrid(xs[0][xi:xi + N], xs[1][yi:yi + N], xs[2], indexing="ij")
                    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).contiguous()
                    val = self.sdf_network.sdf_samples(RaySamples(pts=pts))
                    u[xi:xi + N, yi:yi + N, :] = -val.reshape(xx.shape)
        return u
