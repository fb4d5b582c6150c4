"""Stage-3 material / illumination model with the reference's names (models/inverRender.py), device tensors throughout.

  EnvmapMaterialNetwork   128 direct-light spherical Gaussians (SGs), the BRDF auto-encoder (63 -> 512 x 4 -> 32 -> 128 x 2 -> 4:
                          diffuse albedo + roughness), the specular-albedo MLP net_cs; state_dict keys as the reference
  render_with_all_sg      direct light with per-lobe visibility + the 24 indirect SGs of stage 2, tone mapping
  render_with_sg          SG BRDF (normal-distribution lobe warped to the reflection direction, Fresnel, geometry term) x
                          light SGs x cosine, closed-form hemisphere integrals
  get_diffuse_visibility  per-lobe visibility: 32 directions around every light lobe, the distilled Lvis network evaluated
                          for every (surface point, direction) pair -- 4096 evaluations per point, the hot op of stage 3;
                          runs on the fused HIP kernel fneus_lvis_visibility (fneus/ops.py lvis_visibility)
The SG algebra is element-wise work on [points, 128, 3] tensors (torch ops, differentiable by autograd: the light SGs, the
auto-encoder and net_cs are what stage 3 trains, mateIllu.py:91-95).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from fneus import ops
from models.embedder import get_embedder
from models.fields import _seq_direct, seq_group, DeferredIndirectLight

TINY_NUMBER = 1e-6


def _fused_ok(x):
    return x.is_cuda and x.dtype == torch.float32


def linear_to_srgb(linear):
    """math_utils.py:138-144 (one launch on the GPU: fneus_srgb_fwd / _bwd)"""
    if _fused_ok(linear):
        return ops.srgb(linear)
    eps = torch.finfo(torch.float32).eps
    return torch.where(linear <= 0.0031308, 323.0 / 25.0 * linear,
                       (211.0 * torch.clamp(linear, min=eps) ** (5.0 / 12.0) - 11.0) / 200.0)


def srgb_to_linear(srgb):
    """math_utils.py:147-152"""
    if _fused_ok(srgb):
        return ops.srgb(srgb, to_linear=True)
    eps = torch.finfo(torch.float32).eps
    return torch.where(srgb <= 0.04045, 25.0 / 323.0 * srgb, torch.clamp((200.0 * srgb + 11.0) / 211.0, min=eps) ** (12.0 / 5.0))


def tonemap_clip(linear):
    """torch.clip(tonemap_img(linear), 0, 1) as the reference writes it everywhere (inverRender.py:567-598): one launch"""
    if _fused_ok(linear):
        return ops.srgb(linear, clip=True)
    return torch.clip(linear_to_srgb(linear), 0.0, 1.0)


tonemap_img = linear_to_srgb          # inverRender.py:13-18 with mode = 'dtu'


def norm_axis(x):
    return x / (torch.norm(x, dim=-1, keepdim=True) + TINY_NUMBER)


def compute_energy(lgtSGs):
    """inverRender.py:61-65"""
    lam, mu = torch.abs(lgtSGs[:, 3:4]), torch.abs(lgtSGs[:, 4:])
    return mu * 2.0 * np.pi / lam * (1.0 - torch.exp(-2.0 * lam))


def fibonacci_sphere(samples=1):
    """inverRender.py:68-82"""
    i = np.arange(samples, dtype=np.float64)
    y = 1 - (i / float(samples - 1)) * 2
    radius = np.sqrt(1 - y * y)
    theta = np.pi * (3.0 - np.sqrt(5.0)) * i
    return np.stack([np.cos(theta) * radius, y, np.sin(theta) * radius], axis=1)


def render_envmap_sg(lgtSGs, viewdirs):
    """inverRender.py:36-54: radiance of [M,7] SGs towards [...,3] directions"""
    v = viewdirs.to(lgtSGs.device)[..., None, :]
    lobes = lgtSGs[..., :3] / torch.norm(lgtSGs[..., :3], dim=-1, keepdim=True)
    lam, mu = torch.abs(lgtSGs[..., 3:4]), torch.abs(lgtSGs[..., -3:])
    return (mu * torch.exp(lam * ((v * lobes).sum(-1, keepdim=True) - 1.0))).sum(dim=-2)


def compute_envmap(lgtSGs, H, W, upper_hemi=False):
    """inverRender.py:20-34: latitude-longitude map of the light SGs (Blender convention)"""
    phi, theta = torch.meshgrid(torch.linspace(0.0, np.pi / 2.0 if upper_hemi else np.pi, H, device=lgtSGs.device),
                                torch.linspace(np.pi, -np.pi, W, device=lgtSGs.device), indexing="ij")
    dirs = torch.stack([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], dim=-1)
    return render_envmap_sg(lgtSGs, dirs).reshape(H, W, 3)


def lambda_trick(lobe1, lambda1, mu1, lobe2, lambda2, mu2):
    """product of two SGs for lambda1 << lambda2 (inverRender.py:83-103)"""
    ratio = lambda1 / (lambda2 + TINY_NUMBER)
    lobe1, lobe2 = norm_axis(lobe1), norm_axis(lobe2)
    dot = torch.sum(lobe1 * lobe2, dim=-1, keepdim=True)
    tmp = torch.min(torch.sqrt(ratio * ratio + 1.0 + 2.0 * ratio * dot + TINY_NUMBER), ratio + 1.0)
    lobes = ratio / (tmp + TINY_NUMBER) * lobe1 + 1.0 / (tmp + TINY_NUMBER) * lobe2
    return lobes, lambda2 * tmp, mu1 * mu2 * torch.exp(lambda2 * (tmp - ratio - 1.0))


def hemisphere_int(lambda_val, cos_beta):
    """integral of an SG over the hemisphere around the normal, fitted closed form (inverRender.py:106-125)"""
    lam = torch.clamp(lambda_val, min=TINY_NUMBER)
    inv = 1.0 / (lam + TINY_NUMBER)
    t = torch.sqrt(lam + TINY_NUMBER) * (1.6988 + 10.8438 * inv) / (1.0 + 6.2201 * inv + 10.2415 * inv * inv + TINY_NUMBER)
    inv_a = torch.exp(-t)
    upper = (cos_beta >= 0).float()
    inv_b = torch.exp(-t * torch.clamp(cos_beta, min=0.0))
    s1 = (1.0 - inv_a * inv_b) / (1.0 - inv_a + inv_b - inv_a * inv_b + TINY_NUMBER)
    b = torch.exp(t * torch.clamp(cos_beta, max=0.0))
    s2 = (b - inv_a) / ((1.0 - inv_a) * (b + 1.0) + TINY_NUMBER)
    s = upper * s1 + (1.0 - upper) * s2
    below = 2.0 * np.pi / lam * (torch.exp(-lam) - torch.exp(-2.0 * lam))
    above = 2.0 * np.pi / lam * (1.0 - torch.exp(-lam))
    return below * (1.0 - s) + above * s


def integrate_rgb(normal, final_lobes, final_lambdas, final_mus):
    """sum over lobes of integral(SG x clamped cosine) with the cosine as an SG minus a constant (inverRender.py:264-283)"""
    mu_cos, lambda_cos, alpha_cos = 32.7080, 0.0315, 31.7003
    lobe_p, lambda_p, mu_p = lambda_trick(normal, lambda_cos, mu_cos, final_lobes, final_lambdas, final_mus)
    dot1 = torch.clamp(torch.sum(lobe_p * normal, dim=-1, keepdim=True), min=0.0)
    dot2 = torch.clamp(torch.sum(final_lobes * normal, dim=-1, keepdim=True), min=0.0)
    rgb = mu_p * hemisphere_int(lambda_p, dot1) - final_mus * alpha_cos * hemisphere_int(final_lambdas, dot2)
    return torch.clamp(rgb.sum(dim=-2), min=0.0, max=1.0)


def visibility_sample_dirs(lgtSGLobes, lgtSGLambdas, nsamp, u_theta=None, u_phi=None, lgt_sgs=None):
    """the direction set of get_diffuse_visibility (inverRender.py:133-161): nsamp directions around every light lobe inside
    a cone whose opening follows the lobe's sharpness -> dirs [M, nsamp, 3], weights exp(lambda (d . axis - 1)) [M, nsamp].
    lgt_sgs [M, 7]: the light-SG table the lobes and sharpnesses come from (axis sg[:3] / (|sg[:3]| + 1e-6), sharpness |sg[3]|:
    render_with_all_sg) -- on the GPU they are then taken inside the launch and lgtSGLobes / lgtSGLambdas may be None"""
    if lgt_sgs is not None and lgt_sgs.is_cuda and lgt_sgs.dtype == torch.float32 and lgt_sgs.shape[-1] == 7:
        M, dev = lgt_sgs.shape[0], lgt_sgs.device
        u_theta = torch.rand(M, nsamp, device=dev) if u_theta is None else u_theta
        u_phi = torch.rand(M, nsamp, device=dev) if u_phi is None else u_phi
        return ops.vis_sample_dirs_sgs(lgt_sgs.detach().contiguous(), u_theta.float().contiguous(), u_phi.float().contiguous())
    if lgtSGLobes is None:
        lgtSGLobes = lgt_sgs[:, :3] / (torch.norm(lgt_sgs[:, :3], dim=-1, keepdim=True) + TINY_NUMBER)
        lgtSGLambdas = torch.abs(lgt_sgs[:, 3:4])
    M = lgtSGLobes.shape[0]
    dev = lgtSGLobes.device
    if lgtSGLobes.is_cuda and lgtSGLobes.dtype == torch.float32:       # one launch instead of ~45 element-wise ones
        u_theta = torch.rand(M, nsamp, device=dev) if u_theta is None else u_theta
        u_phi = torch.rand(M, nsamp, device=dev) if u_phi is None else u_phi
        return ops.vis_sample_dirs(lgtSGLobes.detach().float().contiguous(), lgtSGLambdas.detach().float().reshape(-1).contiguous(),
                                   u_theta.float().contiguous(), u_phi.float().contiguous())
    axis = norm_axis(lgtSGLobes.detach()[:, None, :])
    lam = lgtSGLambdas.detach()[:, None, :]
    z_axis = torch.zeros_like(axis)
    z_axis[:, :, 2] = 1
    U = norm_axis(torch.linalg.cross(z_axis, axis, dim=-1))
    V = norm_axis(torch.linalg.cross(axis, U, dim=-1))
    sharp = lam[:, :, 0]
    phi_range = torch.arccos((-1.95 * sharp.min()) / sharp + 1)
    if u_theta is None:
        u_theta = torch.rand(M, nsamp, device=dev)
    if u_phi is None:
        u_phi = torch.rand(M, nsamp, device=dev)
    th, ph = (u_theta * 2 * np.pi)[..., None], (u_phi * phi_range)[..., None]
    dirs = U * torch.cos(th) * torch.sin(ph) + V * torch.sin(th) * torch.sin(ph) + axis * torch.cos(ph)
    w = torch.exp(lam * (torch.sum(dirs * axis, dim=-1, keepdim=True) - 1.0))[..., 0]
    return dirs, w


def get_diffuse_visibility(points, normals, VisModel, lgtSGLobes, lgtSGLambdas, nsamp=8, u_theta=None, u_phi=None,
                           point_mask=None, lgt_sgs=None):
    """inverRender.py:128-192 -> [n_lobe, n_points], detached.  VisModel: models.fields.Lvis.  lgt_sgs: see
    visibility_sample_dirs"""
    from fneus import ops
    with torch.no_grad():
        dirs, w = visibility_sample_dirs(lgtSGLobes, lgtSGLambdas, nsamp, u_theta, u_phi, lgt_sgs=lgt_sgs)
        return VisModel.visibility(points.detach().float().contiguous(), normals.detach().float().contiguous(),
                                   dirs.contiguous(), w.contiguous(), point_mask)


def render_with_sg(points, normal, viewdirs, lgtSGs, specular_reflectance, specular_albedo, roughness, diffuse_albedo,
                   gt_specular_linear=None, comp_vis=True, lvis_network=None, u_theta=None, u_phi=None):
    """inverRender.py:314-449; lgtSGs [n, M, 7]"""
    n, M = lgtSGs.shape[0], lgtSGs.shape[1]
    lobes = lgtSGs[..., :3] / (torch.norm(lgtSGs[..., :3], dim=-1, keepdim=True) + TINY_NUMBER)
    lambdas = torch.abs(lgtSGs[..., 3:4])
    mus0 = torch.abs(lgtSGs[..., -3:])
    nrm = normal[:, None, :].expand(n, M, 3)
    view = viewdirs[:, None, :].expand(n, M, 3).detach()
    inv_r4 = 2.0 / (roughness * roughness * roughness * roughness)
    brdf_lambda = inv_r4[:, None, :].expand(n, M, 1)
    brdf_mu = (inv_r4 / np.pi).expand(n, 3)[:, None, :].expand(n, M, 3)
    v_dot_lobe = torch.clamp(torch.sum(nrm * view, dim=-1, keepdim=True), min=0.0)
    warp_lobes = 2 * v_dot_lobe * nrm - view
    warp_lobes = warp_lobes / (torch.norm(warp_lobes, dim=-1, keepdim=True) + TINY_NUMBER)
    warp_lambdas = brdf_lambda / (4 * v_dot_lobe + TINY_NUMBER)
    half = warp_lobes + view
    half = half / (torch.norm(half, dim=-1, keepdim=True) + TINY_NUMBER)
    v_dot_h = torch.clamp(torch.sum(view * half, dim=-1, keepdim=True), min=0.0)
    f0 = specular_reflectance[:, None, :].expand(n, M, 3)
    fresnel = f0 + (1.0 - f0) * torch.pow(2.0, -(5.55473 * v_dot_h + 6.8316) * v_dot_h)
    dot1 = torch.clamp(torch.sum(warp_lobes * nrm, dim=-1, keepdim=True), min=0.0)
    dot2 = torch.clamp(torch.sum(view * nrm, dim=-1, keepdim=True), min=0.0)
    k = ((roughness + 1.0) * (roughness + 1.0) / 8.0)[:, None, :].expand(n, M, 1)
    G = dot1 / (dot1 * (1 - k) + k + TINY_NUMBER) * (dot2 / (dot2 * (1 - k) + k + TINY_NUMBER))
    Moi = fresnel * G / (4 * dot1 * dot2 + TINY_NUMBER)
    warp_mus = specular_albedo[:, None, :] * brdf_mu * Moi
    vis_shadow = torch.zeros(n, 3, device=points.device)
    if comp_vis:
        light_vis = get_diffuse_visibility(points, normal, lvis_network, lobes[0], lambdas[0], nsamp=32, u_theta=u_theta,
                                           u_phi=u_phi)
        light_vis = light_vis.permute(1, 0)[..., None].expand(n, M, 3)
        mus = mus0 * light_vis
        vis_shadow = torch.mean(light_vis, dim=1)
    else:
        mus = mus0
    fl, fla, fmu = lambda_trick(lobes, lambdas, mus, warp_lobes, warp_lambdas, warp_mus)
    specular_linear = integrate_rgb(nrm, fl, fla, fmu)
    diffuse = (diffuse_albedo / np.pi)[:, None, :].expand(n, M, 3)
    diffuse_linear = integrate_rgb(nrm, lobes, lambdas, mus * diffuse)
    return {"specular_loss": 0, "diffuse_loss": 0, "env_rgb": torch.clamp(specular_linear + diffuse_linear, 0.0, 1.0),
            "diffuse_rgb": tonemap_clip(diffuse_linear),
            "specular_rgb": tonemap_clip(specular_linear), "lvis_mean": vis_shadow}


FUSED_SG = True      # stage 3 on fneus_sg_render_fwd / _bwd (one launch each way); False: the element-wise torch formulation below


def _render_with_all_sg_fused(points, normal, viewdirs, lgtSGs, f0: float, specular_albedo, roughness, diffuse_albedo,
                              lvis_network, indir_lgtSGs, u_theta, u_phi, point_mask=None, want=None, heads=None):
    """render_with_all_sg on the fused kernels: visibility (fneus_lvis_visibility), then every (point, lobe) pair of the 128
    direct and the 24 indirect SGs in one launch; the clamps of integrate_rgb (:277), of render_with_sg (:440) and the tone
    mapping (:306-309) are element-wise ops on [n, 3] tensors.  heads = (brdf [n,4], cs [n,1], direct_lgt): the material as the
    two MLP heads hand it over (fneus.autograd.SgRenderHeadsFn) -- roughness / albedos are then not read here"""
    from fneus.autograd import SgRenderFn, SgRenderHeadsFn
    vis = get_diffuse_visibility(points, normal, lvis_network, None, None, nsamp=32, u_theta=u_theta, u_phi=u_phi,
                                 point_mask=point_mask, lgt_sgs=lgtSGs)           # [M, n], detached
    if heads is not None:
        sums = SgRenderHeadsFn.apply(lgtSGs, heads[0], heads[1], normal, viewdirs, vis, indir_lgtSGs, f0, bool(heads[2]))
    else:
        mat = torch.cat([roughness, diffuse_albedo, specular_albedo], dim=-1)     # [n, 7]
        sums = SgRenderFn.apply(lgtSGs, mat, normal, viewdirs, vis, indir_lgtSGs, f0)
    if want is not None and want <= {"rgb"}:           # the training step: the colour alone, clamps and tone mapping in one launch
        from fneus.autograd import SgCombineFn
        return {"specular_loss": 0, "diffuse_loss": 0, "rgb": SgCombineFn.apply(sums, indir_lgtSGs is not None)}
    # (one clamp and one unbind for the four sums: four slices + clamps are 8 launches forward and ~25 backward -- every slice's
    # backward is a zero fill, a copy and an add on the [n, 4, 3] gradient)
    spec_d, diff_d, spec_i, diff_i = torch.clamp(sums, 0.0, 1.0).unbind(1)
    env = torch.clamp(spec_d + diff_d, 0.0, 1.0)
    indir = torch.clamp(spec_i + diff_i, 0.0, 1.0) if indir_lgtSGs is not None else torch.zeros_like(points)
    ret = {"specular_loss": 0, "diffuse_loss": 0}
    for k, make in (("diffuse_rgb", lambda: tonemap_clip(diff_d)), ("specular_rgb", lambda: tonemap_clip(spec_d)),
                    ("lvis_mean", lambda: vis.mean(dim=0)[:, None].expand(-1, 3)), ("rgb", lambda: tonemap_clip(env + indir)),
                    ("indir_rgb", lambda: tonemap_clip(indir)), ("env_rgb", lambda: tonemap_clip(env))):
        if want is None or k in want:        # (a training step reads `rgb` alone: each of the others is a launch or two)
            ret[k] = make()
    return ret


def _fused_sg_applies(points, lvis_network, specular_reflectance_value, indir_lgtSGs):
    return (FUSED_SG and points.is_cuda and lvis_network is not None and specular_reflectance_value is not None
            and (indir_lgtSGs is None or not indir_lgtSGs.requires_grad))


def render_with_all_sg(points, normal, viewdirs, lgtSGs, specular_reflectance, specular_albedo, roughness, diffuse_albedo,
                       gt_specular_linear=None, lvis_network=None, indir_lgtSGs=None, u_theta=None, u_phi=None,
                       specular_reflectance_value=None, point_mask=None, want=None, heads=None):
    """inverRender.py:286-311: direct light (with visibility) + indirect light, tone mapped.  point_mask [n] bool (fixed-shape
    step): rows marked False are placeholders whose results the caller discards -- their visibility is not evaluated.
    want: set of result keys the caller reads (None = all); the others may be left out"""
    if _fused_sg_applies(points, lvis_network, specular_reflectance_value, indir_lgtSGs):
        return _render_with_all_sg_fused(points, normal, viewdirs, lgtSGs, float(specular_reflectance_value), specular_albedo,
                                         roughness, diffuse_albedo, lvis_network, indir_lgtSGs, u_theta, u_phi, point_mask, want,
                                         heads=heads)
    n = normal.shape[0]
    ret = render_with_sg(points, normal, viewdirs, lgtSGs[None].expand(n, -1, -1), specular_reflectance, specular_albedo,
                         roughness, diffuse_albedo, gt_specular_linear, lvis_network=lvis_network, u_theta=u_theta, u_phi=u_phi)
    indir_rgb = torch.zeros_like(points)
    if indir_lgtSGs is not None:
        indir_rgb = render_with_sg(points, normal, viewdirs, indir_lgtSGs, specular_reflectance, specular_albedo, roughness,
                                   diffuse_albedo, gt_specular_linear, comp_vis=False)["env_rgb"]
    env_rgb = ret["env_rgb"]
    ret.update({"rgb": tonemap_clip(env_rgb + indir_rgb),
                "indir_rgb": tonemap_clip(indir_rgb), "env_rgb": tonemap_clip(env_rgb)})
    return ret


def _mlp(dims, act):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(act)
    return layers


class EnvmapMaterialNetwork(nn.Module):
    """inverRender.py:451-628.  Parameters: lgtSGs [128,7], brdf_encoder_layer.{0,..,8}, brdf_decoder_layer.{0,2,4},
    net_cs.{0,..,8} (first layer an explicit Linear(90, 256) where the reference uses LazyLinear)."""

    def __init__(self, num_lgt_sgs=128, specular_albedo=0.02):
        super().__init__()
        self.numLgtSGs = num_lgt_sgs
        self.embed_view_fn, ch_view = get_embedder(4)
        self.embed_pts_fn, ch_pts = get_embedder(10)
        self.brdf_embed_fn, brdf_in = get_embedder(10)
        self.latent_dim = 32
        self.actv_fn = nn.LeakyReLU(0.2)
        self.brdf_encoder_layer = nn.Sequential(*_mlp([brdf_in, 512, 512, 512, 512, self.latent_dim], self.actv_fn))
        self.brdf_decoder_layer = nn.Sequential(*_mlp([self.latent_dim, 128, 128, 4], self.actv_fn))
        self.net_cs = nn.Sequential(*_mlp([ch_pts + ch_view, 256, 256, 256, 256, 1], nn.LeakyReLU(0.2)), nn.Sigmoid())
        # (a plain tensor attribute in the reference: not part of its state_dict -- a non-persistent buffer moves with .to())
        self.register_buffer("specular_reflectance", torch.full([1, 1], float(specular_albedo)), persistent=False)
        self.specular_reflectance_value = float(specular_albedo)      # the same as a host float for the fused kernels
        # light SGs: grey amplitudes, sharpness 10 + 20 |N|, energy normalised, lobes on two Fibonacci spheres (:509-525)
        sg = torch.randn(num_lgt_sgs, 7)
        sg[:, -2:] = sg[:, -3:-2].expand(-1, 2)
        sg[:, 3:4] = 10.0 + torch.abs(sg[:, 3:4] * 20.0)
        sg[:, 4:] = torch.abs(sg[:, 4:]) / torch.sum(compute_energy(sg), dim=0, keepdim=True) * 2.0 * np.pi * 0.8
        lobes = torch.from_numpy(fibonacci_sphere(num_lgt_sgs // 2).astype(np.float32))
        sg[: num_lgt_sgs // 2, :3] = lobes
        sg[num_lgt_sgs // 2:, :3] = lobes
        self.lgtSGs = nn.Parameter(sg, requires_grad=True)
        self.envmap = None
        self.resolves_deferred_indirect = True      # forward() accepts IndirectLight.deferred(points) as `indiLgt`
        self.stat_reduce = None      # set by the data-parallel trainer: sums the latent-sparsity statistics over the ranks

    def kl_divergence(self, rho, rho_hat, point_mask=None, activated=False):
        """inverRender.py:609-612.  point_mask [n] bool: the mean runs over the marked points only (the fixed-shape stage-3
        step evaluates every ray and marks the ones that hit); without a marked point the term is 0.  activated: `rho_hat` is
        sigmoid(latent) already"""
        red = getattr(self, "stat_reduce", None)
        if rho_hat.is_cuda and rho_hat.dim() == 2 and rho_hat.shape[1] == 32 and rho_hat.dtype == torch.float32 and \
                (red is None or point_mask is None):
            from fneus.autograd import LatentKlFn                   # one launch forward, one backward (~40 element-wise ones)
            return LatentKlFn.apply(rho_hat, point_mask, float(rho), bool(activated))
        act = rho_hat if activated else torch.sigmoid(rho_hat)
        if point_mask is None:
            rho_hat = torch.mean(act, 0)
            return torch.mean(rho * torch.log(rho / rho_hat) + (1 - rho) * torch.log((1 - rho) / (1 - rho_hat)))
        w = point_mask.to(act.dtype)[:, None]
        cnt = w.sum()
        total = (act * w).sum(0)
        red = getattr(self, "stat_reduce", None)
        if red is not None:         # data parallel: the mean runs over the hit points of ALL ranks; the other ranks' part is a
            stats = red(torch.cat([total.detach(), cnt.reshape(1)]))        # constant here (their gradients are theirs)
            total, cnt = total + (stats[:-1] - total.detach()), stats[-1]
        some = cnt > 0
        rho_hat = torch.where(some, total / cnt.clamp(min=1.0), torch.full_like(act[0], rho))
        kl = torch.mean(rho * torch.log(rho / rho_hat) + (1 - rho) * torch.log((1 - rho) / (1 - rho_hat)))
        return torch.where(some, kl, torch.zeros_like(kl))

    def forward(self, points, ray_dirs, n, f, gt_specular_linear, indiLgt, lvis_network, u_theta=None, u_phi=None,
                point_mask=None, want=None):
        """want: set of result keys the caller reads (None = all, the reference's dict); the losses are always returned"""
        from fneus import ops
        fused_in = (points.is_cuda and all(t.dtype == torch.float32 and not t.requires_grad for t in (points, ray_dirs, n))
                    and points.dim() == 2)
        if fused_in:       # normalisations, reflected direction, the three encodings and the concatenation in one launch
            n, view_dirs, brdf_in, cs_in = ops.material_inputs(points.contiguous(), ray_dirs.contiguous(), n.contiguous())
        else:
            n = n / (torch.norm(n, dim=-1, keepdim=True) + TINY_NUMBER)
            ray_dirs = ray_dirs / (torch.norm(ray_dirs, dim=-1, keepdim=True) + TINY_NUMBER)
            view_dirs = -ray_dirs
            ref_dirs = 2.0 * torch.sum(view_dirs * n, dim=-1, keepdim=True) * n - view_dirs
            brdf_in = self.brdf_embed_fn(points)
            cs_in = torch.cat([self.embed_pts_fn(points), self.embed_view_fn(ref_dirs)], dim=-1)
        # The encoder and net_cs have their inputs now: layer by layer in the same launches (models/fields.py seq_group; inside
        # Stage3Trainer's own steps the parameter gradients go straight into their persistent buffers).  The two sigmoids of :555-556
        # are the last activations of the encoder and the decoder there (the latent code is read through its sigmoid only).
        items = [(self.brdf_encoder_layer, brdf_in, self, ops.ACT_SIGMOID), (self.net_cs, cs_in, self)]
        if isinstance(indiLgt, DeferredIndirectLight):        # the frozen IndirectLight of the same points: the same launches
            same = fused_in and indiLgt.pts.data_ptr() == points.data_ptr() and indiLgt.pts.shape == points.shape
            items.append((indiLgt.net.indi, brdf_in if same else indiLgt.net.embedview_fn_pts(indiLgt.pts), indiLgt.net))
        outs = seq_group(items)
        act_latent, cs = outs[0], outs[1]
        if isinstance(indiLgt, DeferredIndirectLight):
            indiLgt = indiLgt.net.sgs_from_raw(outs[2])
        brdf = seq_group([(self.brdf_decoder_layer, act_latent, self, ops.ACT_SIGMOID)])[0]
        loss = 0.01 * self.kl_divergence(0.05, act_latent, point_mask, activated=True)
        # A training step reads the rendered colour alone: the SG kernels then take the two heads' outputs as they are (roughness =
        # 0.9 raw + 0.09 and the three equal specular channels inside the launch) and write the light table's gradient straight into
        # its persistent buffer; the [n, 7] material table and `roughness` exist only for callers that ask for them.
        lean = (want is not None and "roughness" not in want and _fused_sg_applies(points, lvis_network, self.specular_reflectance_value, indiLgt)
                and brdf.dtype == torch.float32)
        heads = None
        if lean:
            direct_lgt = (getattr(self, "direct_grads", False) and torch.is_grad_enabled() and self.lgtSGs.requires_grad
                          and self.lgtSGs.grad is not None and self.lgtSGs.grad.is_contiguous())
            heads = (brdf, cs, direct_lgt)
            diffuse_albedo = roughness = specular_albedo = None
        else:
            diffuse_albedo, rough_raw = torch.split(brdf, [3, 1], dim=-1)      # (split: its backward is one concatenation)
            roughness = rough_raw * 0.9 + 0.09
            specular_albedo = cs.expand(-1, 3)
        ret = render_with_all_sg(points, n, view_dirs, self.lgtSGs, self.specular_reflectance, specular_albedo, roughness,
                                 diffuse_albedo, gt_specular_linear, lvis_network=lvis_network, indir_lgtSGs=indiLgt,
                                 u_theta=u_theta, u_phi=u_phi, specular_reflectance_value=self.specular_reflectance_value,
                                 point_mask=point_mask, want=want, heads=heads)
        ret.update({"encoder_loss": loss, "smooth_loss": 0.0})
        if not lean:
            ret["roughness"] = roughness
        if want is None or "diffuse_albedo" in want:
            ret["diffuse_albedo"] = tonemap_clip(brdf[:, :3] if diffuse_albedo is None else diffuse_albedo)
        if want is None or "specular_albedo" in want:
            ret["specular_albedo"] = tonemap_clip(cs.expand(-1, 3) if specular_albedo is None else specular_albedo)
        return ret

    def get_light(self):
        return compute_envmap(lgtSGs=self.lgtSGs, H=256, W=512)
