"""CPU oracle for the Factored-NeuS volume-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path (the package under
``factored-neus_amd/``) may import this module.  It is used by ``tests/``, by
``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py`` as the
checker / the timed CPU port, never as the thing shipped.

It is a clean-room restatement (own structure, explicit maths) of the reference
algorithm; every function cites the reference ``file:line`` it follows
(paths relative to the upstream repository root).  Parity pinning: the reference
has no tests/golden vectors for this path, so the pins are the
fixtures under ``tests/golden/`` which ``tests/golden/gen_golden.py`` produced by
importing the reference itself in the build container; ``tests/test_oracle_golden.py``
checks this module against them.

All functions are dtype/device generic (fp32 for parity work, fp64 for derivations).
Networks are described by plain dicts of *effective* weights so the same code can be
driven from reference ``state_dict``s:

    sdf_params   = {"W": [W0..W8], "b": [b0..b8], "scale": float}
    color_params = {"W": [W0..W4], "b": [b0..b4]}
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

SOFTPLUS_BETA = 100.0        # fields.py:72  nn.Softplus(beta=100)
SOFTPLUS_THRESHOLD = 20.0    # torch default threshold (beta*x > 20 -> identity)


# --------------------------------------------------------------------------------------
# embedder.py:6-51  positional encoding
# --------------------------------------------------------------------------------------
def embed(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]  (embedder.py:23-36).

    freq_bands = 2**linspace(0, L-1, L) are exact powers of two (embedder.py:23).
    """
    outs = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, dim=-1)


def embed_jacobian_apply_T(x: torch.Tensor, q: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """n = J^T q where J = d embed(x) / dx  (diagonal per coordinate).  SURVEY Appendix A."""
    d = x.shape[-1]
    n = q[..., :d].clone()
    for k in range(n_freqs):
        f = float(2 ** k)
        qs = q[..., d * (1 + 2 * k): d * (2 + 2 * k)]
        qc = q[..., d * (2 + 2 * k): d * (3 + 2 * k)]
        n = n + f * torch.cos(x * f) * qs - f * torch.sin(x * f) * qc
    return n


# --------------------------------------------------------------------------------------
# weight norm:  nn.utils.weight_norm(lin) with dim=0  (fields.py:67-68, 139-140)
# --------------------------------------------------------------------------------------
def fold_weight_norm(weight_g: torch.Tensor, weight_v: torch.Tensor) -> torch.Tensor:
    """W = g * v / ||v||_row  (torch._weight_norm, dim=0: one norm per output row)."""
    return weight_v * (weight_g / weight_v.norm(dim=1, keepdim=True))


def sdf_params_from_state_dict(sd: Dict[str, torch.Tensor], scale: float = 1.0, n_lin: int = 9) -> dict:
    """Fold the reference SDFNetwork state_dict (keys lin{l}.weight_g/weight_v/bias, fields.py:67-70)."""
    Ws, bs = [], []
    for l in range(n_lin):
        if f"lin{l}.weight_g" in sd:
            Ws.append(fold_weight_norm(sd[f"lin{l}.weight_g"], sd[f"lin{l}.weight_v"]))
        else:
            Ws.append(sd[f"lin{l}.weight"])
        bs.append(sd[f"lin{l}.bias"])
    return {"W": Ws, "b": bs, "scale": scale}


def color_params_from_state_dict(sd: Dict[str, torch.Tensor], n_lin: int = 5) -> dict:
    Ws, bs = [], []
    for l in range(n_lin):
        if f"lin{l}.weight_g" in sd:
            Ws.append(fold_weight_norm(sd[f"lin{l}.weight_g"], sd[f"lin{l}.weight_v"]))
        else:
            Ws.append(sd[f"lin{l}.weight"])
        bs.append(sd[f"lin{l}.bias"])
    return {"W": Ws, "b": bs}


# --------------------------------------------------------------------------------------
# fields.py:74-111  SDFNetwork forward / sdf / gradient
# --------------------------------------------------------------------------------------
def softplus100(z: torch.Tensor) -> torch.Tensor:
    return F.softplus(z, beta=SOFTPLUS_BETA, threshold=SOFTPLUS_THRESHOLD)


def softplus100_d1(z: torch.Tensor) -> torch.Tensor:
    """d softplus / dz = sigmoid(beta z); exactly 1 in the linear region (beta z > 20)."""
    s = torch.sigmoid(SOFTPLUS_BETA * z)
    return torch.where(SOFTPLUS_BETA * z > SOFTPLUS_THRESHOLD, torch.ones_like(s), s)


def softplus100_d2(z: torch.Tensor) -> torch.Tensor:
    s = torch.sigmoid(SOFTPLUS_BETA * z)
    d2 = SOFTPLUS_BETA * s * (1.0 - s)
    return torch.where(SOFTPLUS_BETA * z > SOFTPLUS_THRESHOLD, torch.zeros_like(s), d2)


def sdf_forward(x: torch.Tensor, p: dict, multires: int = 6, skip_in: Sequence[int] = (4,),
                keep: bool = False):
    """SDFNetwork.forward (fields.py:74-91).  x:[M,3] -> [M, 1+feat].

    With keep=True also returns the per-layer inputs u_l and pre-activations z_l.
    """
    scale = p.get("scale", 1.0)
    Ws, bs = p["W"], p["b"]
    n_lin = len(Ws)
    h0 = embed(x * scale, multires) if multires > 0 else x * scale
    h = h0
    us, zs = [], []
    for l in range(n_lin):
        if l in skip_in:
            h = torch.cat([h, h0], dim=-1) / math.sqrt(2.0)      # fields.py:83-84
        z = h @ Ws[l].t() + bs[l]                                 # fields.py:86
        us.append(h)
        zs.append(z)
        h = softplus100(z) if l < n_lin - 1 else z                 # fields.py:88-89
    out = torch.cat([h[:, :1] / scale, h[:, 1:]], dim=-1)         # fields.py:91
    if keep:
        return out, us, zs, h0
    return out


def sdf_only(x: torch.Tensor, p: dict, **kw) -> torch.Tensor:
    """SDFNetwork.sdf (fields.py:93-95)."""
    return sdf_forward(x, p, **kw)[:, :1]


def sdf_value_feature_normal(x: torch.Tensor, p: dict, multires: int = 6,
                             skip_in: Sequence[int] = (4,)):
    """sdf, feature and the analytic normal d sdf / d x in one pass.

    Equivalent to SDFNetwork.forward + SDFNetwork.gradient (fields.py:74-111): the
    reference obtains the normal with autograd.grad(create_graph=True); here the same
    derivative is written out as the reverse sweep of SURVEY Appendix A so that the HIP
    kernel's intermediate quantities (a_l, g_hat) have a checker.  Differentiable by torch
    autograd (double backward) because it is built from differentiable torch ops.
    Returns sdf[M,1], feature[M,F], normal[M,3], aux(dict of intermediates).
    """
    scale = p.get("scale", 1.0)
    Ws = p["W"]
    n_lin = len(Ws)
    out, us, zs, h0 = sdf_forward(x, p, multires=multires, skip_in=skip_in, keep=True)
    d0 = h0.shape[-1]
    # reverse sweep: g = d sdf_raw / d u_l
    g = Ws[n_lin - 1][0:1, :].expand(x.shape[0], -1)              # row 0 of the last layer
    q_skip = None
    a_list = [None] * n_lin
    ghat_list = [None] * n_lin
    for l in range(n_lin - 2, -1, -1):
        if (l + 1) in skip_in:
            g = g / math.sqrt(2.0)
            q_skip = g[:, -d0:]
            g = g[:, :-d0]
        ghat_list[l] = g
        a = softplus100_d1(zs[l]) * g
        a_list[l] = a
        g = a @ Ws[l]
    q = g if q_skip is None else g + q_skip
    if multires > 0:
        normal = embed_jacobian_apply_T(x * scale, q, multires)
    else:
        normal = q
    # chain rule for the input scale and the output 1/scale cancel: d(sdf_raw/scale)/dx = scale*J^T q/scale
    aux = {"u": us, "z": zs, "a": a_list, "ghat": ghat_list, "q": q, "h0": h0}
    return out[:, :1], out[:, 1:], normal, aux


def sdf_gradient_autograd(x: torch.Tensor, p: dict, **kw) -> torch.Tensor:
    """SDFNetwork.gradient exactly as the reference does it (fields.py:100-111); checker for the analytic sweep."""
    x = x.detach().clone().requires_grad_(True)
    y = sdf_only(x, p, **kw)
    (g,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)
    return g


# --------------------------------------------------------------------------------------
# fields.py:262-268  SingleVarianceNetwork
# --------------------------------------------------------------------------------------
def inv_s_from_variance(variance: torch.Tensor) -> torch.Tensor:
    """exp(10*variance), clipped like the caller does (fields.py:267-268, renderer.py:245)."""
    return torch.exp(variance * 10.0).clip(1e-6, 1e6)


# --------------------------------------------------------------------------------------
# fields.py:150-175  RenderingNetwork (mode 'idr')
# --------------------------------------------------------------------------------------
def color_forward(points, normals, view_dirs, features, p: dict, multires_view: int = 4,
                  squeeze_out: bool = True, keep: bool = False):
    v = embed(view_dirs, multires_view) if multires_view > 0 else view_dirs   # fields.py:151-152
    x = torch.cat([points, v, normals, features], dim=-1)                     # fields.py:157
    Ws, bs = p["W"], p["b"]
    us, zs = [], []
    for l in range(len(Ws)):
        us.append(x)
        x = x @ Ws[l].t() + bs[l]                                             # fields.py:168
        zs.append(x)
        if l < len(Ws) - 1:
            x = torch.relu(x)                                                 # fields.py:170-171
    if squeeze_out:
        x = torch.sigmoid(x)                                                  # fields.py:173-174
    if keep:
        return x, us, zs
    return x


# --------------------------------------------------------------------------------------
# fields.py:233-259  NeRF (background, n_outside > 0)
# --------------------------------------------------------------------------------------
def nerf_forward(input_pts, input_views, sd: Dict[str, torch.Tensor], multires=10, multires_view=4,
                 D=8, skips=(4,)):
    """NeRF.forward with use_viewdirs=True; sd is the reference state_dict (plain nn.Linear)."""
    pe = embed(input_pts, multires)
    ve = embed(input_views, multires_view)
    h = pe
    for i in range(D):
        h = torch.relu(h @ sd[f"pts_linears.{i}.weight"].t() + sd[f"pts_linears.{i}.bias"])
        if i in skips:
            h = torch.cat([pe, h], dim=-1)                                    # fields.py:244-245
    alpha = h @ sd["alpha_linear.weight"].t() + sd["alpha_linear.bias"]
    feat = h @ sd["feature_linear.weight"].t() + sd["feature_linear.bias"]
    h = torch.cat([feat, ve], dim=-1)
    h = torch.relu(h @ sd["views_linears.0.weight"].t() + sd["views_linears.0.bias"])
    rgb = h @ sd["rgb_linear.weight"].t() + sd["rgb_linear.bias"]
    return alpha, rgb


# --------------------------------------------------------------------------------------
# math_utils.py:12-22, 138-144 and fields.py:303-335  RefColor
# --------------------------------------------------------------------------------------
def l2_normalize(x):
    eps = torch.finfo(torch.float32).eps
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, dim=-1, keepdim=True), min=eps))


def reflect(d, n):
    return 2.0 * torch.sum(d * n, dim=-1, keepdim=True) * n - d


def linear_to_srgb(linear):
    eps = torch.finfo(torch.float32).eps
    srgb0 = 323.0 / 25.0 * linear
    srgb1 = (211.0 * torch.clamp(linear, min=eps) ** (5.0 / 12.0) - 11.0) / 200.0
    return torch.where(linear <= 0.0031308, srgb0, srgb1)


def refcolor_forward(pts, feat, dirs, n, sd: Dict[str, torch.Tensor]):
    """RefColor.forward (fields.py:303-335); sd = reference state_dict (net_cd.*, viewdir_mlp.*, net_cs.*)."""
    normals = l2_normalize(n)
    n_enc = embed(n, 4)                                    # raw n is encoded (fields.py:306)
    ref_dirs = reflect(-dirs, normals)
    ref_enc = embed(ref_dirs, 4)
    x = torch.cat([pts, n_enc, feat], dim=-1)
    for i in (0, 2, 4, 6):
        x = torch.relu(x @ sd[f"net_cd.{i}.weight"].t() + sd[f"net_cd.{i}.bias"])
    diffuse = torch.sigmoid(x @ sd["net_cd.8.weight"].t() + sd["net_cd.8.bias"])
    inputs_cs = torch.cat([n, pts, ref_enc, feat], dim=-1)
    x2 = inputs_cs
    for i in range(4):
        x2 = torch.relu(x2 @ sd[f"viewdir_mlp.{i}.weight"].t() + sd[f"viewdir_mlp.{i}.bias"])
        # (the reference's "i % 4 == 0 and i > 0" re-concat never fires for i in 0..3, fields.py:319-320)
    spec = torch.sigmoid(x2 @ sd["net_cs.0.weight"].t() + sd["net_cs.0.bias"]).repeat(1, 3)
    brdf = spec + diffuse
    return {
        "rgb": torch.clip(linear_to_srgb(brdf), 0.0, 1.0),
        "specular_rgb": torch.clip(linear_to_srgb(spec), 0.0, 1.0),
        "diffuse_rgb": torch.clip(linear_to_srgb(diffuse), 0.0, 1.0),
    }


# --------------------------------------------------------------------------------------
# renderer.py:43-77  sample_pdf
# --------------------------------------------------------------------------------------
def sample_pdf_det(bins: torch.Tensor, weights: torch.Tensor, n_new: int, return_bins: bool = False):
    """Deterministic inverse-CDF sampling at u=(k+.5)/n_new  (renderer.py:43-77, det=True).
    return_bins: also the bin index each sample was drawn from (`below`) and the cdf."""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)                 # [B, m]
    u = torch.linspace(0.5 / n_new, 1.0 - 0.5 / n_new, n_new, dtype=bins.dtype, device=bins.device)
    u = u.expand(cdf.shape[0], n_new).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    out = bin_b + t * (bin_a - bin_b)
    return (out, below, cdf) if return_bins else out


def exclusive_transmittance(alpha: torch.Tensor) -> torch.Tensor:
    """T_i = prod_{j<i} (1 - alpha_j + 1e-7)   (renderer.py:140, 183-184, 360)."""
    one = torch.ones_like(alpha[:, :1])
    return torch.cumprod(torch.cat([one, 1.0 - alpha + 1e-7], -1), -1)[:, :-1]


# --------------------------------------------------------------------------------------
# renderer.py:152-205  up_sample / cat_z_vals
# --------------------------------------------------------------------------------------
def up_sample(rays_o, rays_d, z_vals, sdf, n_new: int, inv_s: float, return_bins: bool = False):
    B, m = z_vals.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]
    radius = torch.linalg.norm(pts, dim=-1)
    inside = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
    sdf = sdf.reshape(B, m)
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    prev_z, next_z = z_vals[:, :-1], z_vals[:, 1:]
    mid_sdf = (prev_sdf + next_sdf) * 0.5
    cos = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-5)
    prev_cos = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], -1)
    cos = torch.minimum(prev_cos, cos).clip(-1e3, 0.0) * inside
    dist = next_z - prev_z
    prev_est = mid_sdf - cos * dist * 0.5
    next_est = mid_sdf + cos * dist * 0.5
    prev_cdf = torch.sigmoid(prev_est * inv_s)
    next_cdf = torch.sigmoid(next_est * inv_s)
    alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
    weights = alpha * exclusive_transmittance(alpha)
    return sample_pdf_det(z_vals, weights, n_new, return_bins=return_bins)


def cat_z_vals(rays_o, rays_d, z_vals, new_z, sdf, sdf_fn, last: bool):
    """renderer.py:191-205.  sdf_fn maps [M,3] -> [M,1]."""
    B, m = z_vals.shape
    k = new_z.shape[1]
    z_all = torch.cat([z_vals, new_z], -1)
    z_sorted, index = torch.sort(z_all, dim=-1)
    if not last:
        pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z[..., None]
        new_sdf = sdf_fn(pts.reshape(-1, 3)).reshape(B, k)
        sdf = torch.gather(torch.cat([sdf, new_sdf], -1), 1, index)
    return z_sorted, sdf


def hierarchical_z(rays_o, rays_d, z_vals, sdf_fn, n_importance: int, up_sample_steps: int,
                   trace: Optional[list] = None):
    """The no-grad up-sampling loop of NeuSRenderer.render (renderer.py:425-449)."""
    B, n = z_vals.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]
    sdf = sdf_fn(pts.reshape(-1, 3)).reshape(B, n)
    for i in range(up_sample_steps):
        new_z = up_sample(rays_o, rays_d, z_vals, sdf, n_importance // up_sample_steps, 64 * 2 ** i)
        z_vals, sdf = cat_z_vals(rays_o, rays_d, z_vals, new_z, sdf, sdf_fn,
                                 last=(i + 1 == up_sample_steps))
        if trace is not None:
            trace.append((new_z, z_vals, sdf))
    return z_vals


# --------------------------------------------------------------------------------------
# renderer.py:112-149  render_core_outside
# --------------------------------------------------------------------------------------
def render_core_outside(rays_o, rays_d, z_vals, sample_dist, nerf_fn, background_rgb=None):
    B, n = z_vals.shape
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], sample_dist)], -1)
    mid_z = z_vals + dists * 0.5
    pts = rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]
    dis = torch.linalg.norm(pts, dim=-1, keepdim=True).clip(1.0, 1e10)
    pts4 = torch.cat([pts / dis, 1.0 / dis], dim=-1).reshape(-1, 4)
    dirs = rays_d[:, None, :].expand(B, n, 3).reshape(-1, 3)
    density, rgb = nerf_fn(pts4, dirs)
    rgb = torch.sigmoid(rgb).reshape(B, n, 3)
    alpha = 1.0 - torch.exp(-F.softplus(density.reshape(B, n)) * dists)
    weights = alpha * exclusive_transmittance(alpha)
    color = (weights[..., None] * rgb).sum(1)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - weights.sum(-1, keepdim=True))
    return {"color": color, "sampled_color": rgb, "alpha": alpha, "weights": weights}


# --------------------------------------------------------------------------------------
# renderer.py:208-389  render_core
# --------------------------------------------------------------------------------------
def neus_alpha(sdf, true_cos, dists, inv_s, cos_anneal_ratio):
    """SDF -> alpha of one section (renderer.py:248-268).  All [.,1] or broadcastable."""
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio)
                 + F.relu(-true_cos) * cos_anneal_ratio)
    est_next = sdf + iter_cos * dists * 0.5
    est_prev = sdf - iter_cos * dists * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)
    return alpha, prev_cdf


def render_core(rays_o, rays_d, z_vals, sample_dist, sdf_p, inv_s, color_p, refcolor_sd=None,
                background_alpha=None, background_sampled_color=None, background_rgb=None,
                cos_anneal_ratio=0.0, multires=6, multires_view=4, analytic_normal=True):
    B, n = z_vals.shape
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], sample_dist)], -1)
    mid_z = z_vals + dists * 0.5                                                   # renderer.py:223-226
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(B, n, 3).reshape(-1, 3)

    if analytic_normal:
        sdf, feat, grad, _ = sdf_value_feature_normal(pts, sdf_p, multires=multires)  # renderer.py:238-242
    else:
        out = sdf_forward(pts, sdf_p, multires=multires)
        sdf, feat = out[:, :1], out[:, 1:]
        grad = sdf_gradient_autograd(pts, sdf_p, multires=multires)

    inv_s_col = inv_s.reshape(1, 1).expand(B * n, 1)                                # renderer.py:245-246
    true_cos = (dirs * grad).sum(-1, keepdim=True)                                  # renderer.py:248
    alpha, prev_cdf = neus_alpha(sdf, true_cos, dists.reshape(-1, 1), inv_s_col, cos_anneal_ratio)
    alpha = alpha.reshape(B, n)

    pts_norm = torch.linalg.norm(pts, dim=-1).reshape(B, n)
    inside = (pts_norm < 1.0).to(z_vals.dtype)                                      # renderer.py:270-272
    relax_inside = (pts_norm < 1.2).to(z_vals.dtype)
    inside_ray = inside.sum(-1) > 0.0                                               # renderer.py:274

    sampled_color = color_forward(pts, grad, dirs, feat, color_p,
                                  multires_view=multires_view).reshape(B, n, 3)     # renderer.py:278

    ones3 = torch.ones(B, 3, dtype=z_vals.dtype, device=z_vals.device)
    specular_color, diffuse_color, surface_color = ones3.clone(), ones3.clone(), ones3.clone()

    # first sign change (renderer.py:290-293)
    sdf_bn = sdf.reshape(B, n)
    ramp = torch.arange(n, 0, -1, dtype=z_vals.dtype, device=z_vals.device).reshape(1, n)
    min_val, min_idx = torch.min(torch.sign(sdf_bn) * ramp, dim=-1)
    sdf_mask = (min_val < 0.0) & (min_idx >= 1) & inside_ray
    if refcolor_sd is not None and int(sdf_mask.sum()) > 0:                          # renderer.py:296-343
        rows = torch.nonzero(sdf_mask).squeeze(-1)
        hi_idx = min_idx[rows]
        lo_idx = hi_idx - 1
        flat_lo = rows * n + lo_idx
        flat_hi = rows * n + hi_idx
        sel = torch.stack([flat_lo, flat_hi], dim=1).reshape(-1)                     # low, high interleaved
        ref = refcolor_forward(pts[sel], feat[sel], dirs[sel], grad[sel], refcolor_sd)
        alpha_in = alpha * inside
        w_in = alpha_in * exclusive_transmittance(alpha_in)
        w_lo = w_in.reshape(-1)[flat_lo].unsqueeze(-1) + 1e-5
        w_hi = w_in.reshape(-1)[flat_hi].unsqueeze(-1) + 1e-5

        def blend(v):
            v = v.reshape(-1, 2, 3)
            return (v[:, 0] * w_lo + v[:, 1] * w_hi) / (w_lo + w_hi)

        specular_color = specular_color.index_put((rows,), blend(ref["specular_rgb"]))
        diffuse_color = diffuse_color.index_put((rows,), blend(ref["diffuse_rgb"]))
        surface_color = surface_color.index_put((rows,), blend(ref["rgb"]))

    if background_alpha is not None:                                                 # renderer.py:350-356
        alpha = alpha * inside + background_alpha[:, :n] * (1.0 - inside)
        alpha = torch.cat([alpha, background_alpha[:, n:]], dim=-1)
        sampled_color = sampled_color * inside[..., None] + \
            background_sampled_color[:, :n] * (1.0 - inside)[..., None]
        sampled_color = torch.cat([sampled_color, background_sampled_color[:, n:]], dim=1)

    weights = alpha * exclusive_transmittance(alpha)                                 # renderer.py:360
    weights_sum = weights.sum(-1, keepdim=True)
    color = (sampled_color * weights[..., None]).sum(1)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - weights_sum)

    gnorm = torch.linalg.norm(grad.reshape(B, n, 3), dim=-1)
    gradient_error = (relax_inside * (gnorm - 1.0) ** 2).sum() / (relax_inside.sum() + 1e-5)  # renderer.py:370-372

    return {
        "color": color, "surface_color": surface_color, "sdf_mask": sdf_mask, "sdf": sdf,
        "dists": dists, "gradients": grad.reshape(B, n, 3), "s_val": 1.0 / inv_s_col,
        "mid_z_vals": mid_z, "weights": weights, "cdf": prev_cdf.reshape(B, n),
        "gradient_error": gradient_error, "inside_sphere": inside,
        "specular_color": specular_color, "diffuse_color": diffuse_color,
        "alpha": alpha, "sampled_color": sampled_color, "feature": feat, "min_sdf_idx": min_idx,
    }


# --------------------------------------------------------------------------------------
# renderer.py:391-500  render
# --------------------------------------------------------------------------------------
def initial_z_vals(near, far, n_samples: int, t_rand: Optional[torch.Tensor] = None):
    """renderer.py:394-395, 407-409.  t_rand:[B,1] in [0,1) or None (perturb off)."""
    z = torch.linspace(0.0, 1.0, n_samples, dtype=near.dtype, device=near.device)
    z = near + (far - near) * z[None, :]
    if t_rand is not None:
        z = z + (t_rand - 0.5) * 2.0 / n_samples
    return z


def outside_z_vals(far, n_samples: int, n_outside: int, t_rand_out: Optional[torch.Tensor] = None):
    """renderer.py:397-419."""
    zo = torch.linspace(1e-3, 1.0 - 1.0 / (n_outside + 1.0), n_outside, dtype=far.dtype, device=far.device)
    if t_rand_out is not None:
        mids = 0.5 * (zo[1:] + zo[:-1])
        upper = torch.cat([mids, zo[-1:]], -1)
        lower = torch.cat([zo[:1], mids], -1)
        zo = lower[None, :] + (upper - lower)[None, :] * t_rand_out
    return far / torch.flip(zo, dims=[-1]) + 1.0 / n_samples


def render(rays_o, rays_d, near, far, sdf_p, inv_s, color_p, refcolor_sd=None, nerf_sd=None,
           n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4,
           t_rand=None, t_rand_out=None, background_rgb=None, cos_anneal_ratio=0.0,
           multires=6, multires_view=4, analytic_normal=True, trace: Optional[list] = None,
           z_vals_override: Optional[torch.Tensor] = None):
    """NeuSRenderer.render (renderer.py:391-500).  Randomness is passed in (t_rand / t_rand_out).

    z_vals_override (test hook): skip the hierarchical sampler and use these sorted z instead.  The sampler is
    an ill-conditioned map (inverse CDF of a nearly flat pdf), so stage-wise parity checks feed the checker's
    own z into the stage under test ("teacher forcing")."""
    B = rays_o.shape[0]
    sample_dist = 2.0 / n_samples
    z_vals = initial_z_vals(near, far, n_samples, t_rand)
    z_out = outside_z_vals(far, n_samples, n_outside, t_rand_out) if n_outside > 0 else None
    n = n_samples
    if n_importance > 0:
        if z_vals_override is not None:
            z_vals = z_vals_override
        else:
            with torch.no_grad():
                sdf_fn = lambda q: sdf_only(q, sdf_p, multires=multires)
                z_vals = hierarchical_z(rays_o, rays_d, z_vals, sdf_fn, n_importance, up_sample_steps, trace)
        n = n_samples + n_importance
    bg_alpha = bg_color = None
    if n_outside > 0:
        z_feed, _ = torch.sort(torch.cat([z_vals, z_out], -1), dim=-1)
        nerf_fn = lambda a, b: nerf_forward(a, b, nerf_sd)
        ro = render_core_outside(rays_o, rays_d, z_feed, sample_dist, nerf_fn)
        bg_color, bg_alpha = ro["sampled_color"], ro["alpha"]
    rc = render_core(rays_o, rays_d, z_vals, sample_dist, sdf_p, inv_s, color_p, refcolor_sd,
                     background_alpha=bg_alpha, background_sampled_color=bg_color,
                     background_rgb=background_rgb, cos_anneal_ratio=cos_anneal_ratio,
                     multires=multires, multires_view=multires_view, analytic_normal=analytic_normal)
    weights = rc["weights"]
    return {
        "color_fine": rc["color"], "surface_color": rc["surface_color"], "sdf_mask": rc["sdf_mask"],
        "s_val": rc["s_val"].reshape(B, n).mean(-1, keepdim=True), "cdf_fine": rc["cdf"],
        "weight_sum": weights.sum(-1, keepdim=True), "weight_max": weights.max(-1, keepdim=True)[0],
        "gradients": rc["gradients"], "weights": weights, "gradient_error": rc["gradient_error"],
        "inside_sphere": rc["inside_sphere"], "specular_color": rc["specular_color"],
        "diffuse_color": rc["diffuse_color"],
        # extras for the parity tests (not part of the reference dict)
        "_z_vals": z_vals, "_sdf": rc["sdf"], "_alpha": rc["alpha"], "_sampled_color": rc["sampled_color"],
        "_mid_z_vals": rc["mid_z_vals"], "_feature": rc["feature"], "_min_sdf_idx": rc["min_sdf_idx"],
    }


# --------------------------------------------------------------------------------------
# exp_runner.py:141-177  stage-1 losses
# --------------------------------------------------------------------------------------
def stage1_loss(out: dict, true_rgb, mask_in, igr_weight=0.1, mask_weight=0.1, surface_weight=0.1):
    if mask_weight > 0.0:
        mask = (mask_in > 0.5).to(true_rgb.dtype)
    else:
        mask = torch.ones_like(mask_in)
    mask_sum = mask.sum() + 1e-5
    color_err = (out["color_fine"] - true_rgb) * mask
    color_loss = color_err.abs().sum() / mask_sum
    sm = out["sdf_mask"]
    mask_sdf_sum = mask[sm].sum() + 1e-5
    surf_err = surface_weight * (out["surface_color"][sm] - true_rgb[sm]) * mask[sm]
    surface_loss = surf_err.abs().sum() / mask_sdf_sum
    eik = out["gradient_error"]
    mask_loss = F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    loss = color_loss + surface_loss + eik * igr_weight + mask_loss * mask_weight
    psnr = 20.0 * torch.log10(1.0 / (((out["color_fine"] - true_rgb) ** 2 * mask).sum() / (mask_sum * 3.0)).sqrt())
    return {"loss": loss, "color_loss": color_loss, "surface_loss": surface_loss,
            "eikonal_loss": eik, "mask_loss": mask_loss, "psnr": psnr}


def near_far_from_sphere(rays_o, rays_d):
    """dataset.py:186-192."""
    a = (rays_d ** 2).sum(-1, keepdim=True)
    b = 2.0 * (rays_o * rays_d).sum(-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return mid - 1.0, mid + 1.0


# --------------------------------------------------------------------------------------
# renderer.py:503-564  NeuSRenderer.lvis_mateIllu_render_util: the entry of the stage-2 / stage-3 renderers
# --------------------------------------------------------------------------------------
def lvis_mateIllu_render_util(rays_o, rays_d, near, far, sdf_p, n_samples: int, n_importance: int, up_sample_steps: int = 4):
    """unperturbed hierarchical sampling, SDF at the section mid-points, per-ray inside-sphere mask"""
    sample_dist = 2.0 / n_samples
    z = near + (far - near) * torch.linspace(0.0, 1.0, n_samples, dtype=near.dtype)[None, :]
    sdf_fn = lambda q: sdf_only(q, sdf_p)
    n = n_samples
    if n_importance > 0:
        with torch.no_grad():
            z = hierarchical_z(rays_o, rays_d, z, sdf_fn, n_importance, up_sample_steps)
        n = n_samples + n_importance
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], sample_dist)], -1)
    mid_z = z + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    sdf = sdf_only(pts, sdf_p)
    inside = (torch.linalg.norm(pts, dim=-1).reshape(-1, n) < 1.0).to(z.dtype)
    return {"n_samples": n, "mid_z_vals": mid_z, "sdf": sdf, "inside_sphere_mask": inside.sum(-1) > 0.0}


# --------------------------------------------------------------------------------------
# dataset.py:115-151  Dataset.gen_rays_at / gen_random_rays_at (the pixel -> ray part)
# --------------------------------------------------------------------------------------
def rays_from_pixels(intrinsics_inv, pose, px, py):
    """pixel (x, y) -> K^-1 [x, y, 1] -> normalise -> R v; origin = pose[:3, 3]   (dataset.py:140-146).
    intrinsics_inv, pose: [4,4] of one image; px, py: [...] pixel coordinates (float)."""
    p = torch.stack([px, py, torch.ones_like(px)], -1)
    p = torch.matmul(intrinsics_inv[:3, :3], p[..., None])[..., 0]
    v = p / torch.linalg.norm(p, dim=-1, keepdim=True)
    v = torch.matmul(pose[:3, :3], v[..., None])[..., 0]
    return pose[:3, 3].expand(v.shape), v


def gen_rays_at(intrinsics_inv, pose, H: int, W: int, resolution_level: int = 1):
    """dataset.py:115-131: the full image grid at 1 / resolution_level -> rays_o, rays_v [H/l, W/l, 3]"""
    l = resolution_level
    tx = torch.linspace(0, W - 1, W // l)
    ty = torch.linspace(0, H - 1, H // l)
    px, py = torch.meshgrid(tx, ty, indexing="ij")
    o, v = rays_from_pixels(intrinsics_inv, pose, px, py)
    return o.transpose(0, 1), v.transpose(0, 1)


def gen_random_rays_at(intrinsics_inv, pose, image, mask, px, py):
    """dataset.py:133-151 with the pixel draws given: -> [B, 10] = rays_o, rays_v, colour, mask[:, :1]"""
    o, v = rays_from_pixels(intrinsics_inv, pose, px.float(), py.float())
    return torch.cat([o, v, image[(py, px)], mask[(py, px)][:, :1]], -1)


# --------------------------------------------------------------------------------------
# The per-ray part of render_core on given per-sample fields (checker for the HIP compositing kernel):
# renderer.py:245-274 (alpha, inside), :290-293 (first sign change), :328-332 (inside-sphere weights),
# :360-367 (weights, colour), :370-372 (eikonal sums)
# --------------------------------------------------------------------------------------
def composite_from_fields(rays_o, rays_d, mid_z, dists, sdf, normal, rgb, inv_s, cos_anneal_ratio):
    B, n = mid_z.shape
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(B, n, 3).reshape(-1, 3)
    true_cos = (dirs * normal.reshape(-1, 3)).sum(-1, keepdim=True)
    alpha, prev_cdf = neus_alpha(sdf.reshape(-1, 1), true_cos, dists.reshape(-1, 1),
                                 inv_s.reshape(1, 1).expand(B * n, 1), cos_anneal_ratio)
    alpha = alpha.reshape(B, n)
    pts_norm = torch.linalg.norm(pts, dim=-1).reshape(B, n)
    inside = (pts_norm < 1.0).to(mid_z.dtype)
    relax = (pts_norm < 1.2).to(mid_z.dtype)
    weights = alpha * exclusive_transmittance(alpha)
    color = (rgb.reshape(B, n, 3) * weights[..., None]).sum(1)
    gnorm = torch.linalg.norm(normal.reshape(B, n, 3), dim=-1)
    eik_num = (relax * (gnorm - 1.0) ** 2).sum(-1)
    eik_den = relax.sum(-1)
    ramp = torch.arange(n, 0, -1, dtype=mid_z.dtype).reshape(1, n)
    min_val, min_idx = torch.min(torch.sign(sdf.reshape(B, n)) * ramp, dim=-1)
    sdf_mask = (min_val < 0.0) & (min_idx >= 1) & (inside.sum(-1) > 0.0)
    alpha_in = alpha * inside
    w_in = alpha_in * exclusive_transmittance(alpha_in)
    idx = torch.where(sdf_mask, min_idx, torch.ones_like(min_idx))
    w_lo = torch.gather(w_in, 1, (idx - 1)[:, None])[:, 0] * sdf_mask
    w_hi = torch.gather(w_in, 1, idx[:, None])[:, 0] * sdf_mask
    return {"weights": weights, "color": color, "wsum": weights.sum(-1), "wmax": weights.max(-1)[0],
            "cdf": prev_cdf.reshape(B, n), "inside": inside, "eik_num": eik_num, "eik_den": eik_den,
            "min_idx": min_idx, "sdf_mask": sdf_mask, "wpair": torch.stack([w_lo, w_hi], -1), "alpha": alpha}


# ======================================================================================
# Stage 2 (lvis.py): light visibility / indirect light distillation
# ======================================================================================
def sequential_mlp(x, sd: Dict[str, torch.Tensor], prefix: str, n_lin: int = 5):
    """nn.Sequential(Linear, ReLU, ..., Linear): modules 0, 2, 4, ... are the Linear layers (fields.py:348-359, 387-397)"""
    for i in range(n_lin):
        x = x @ sd[f"{prefix}.{2 * i}.weight"].t() + sd[f"{prefix}.{2 * i}.bias"]
        if i < n_lin - 1:
            x = torch.relu(x)
    return x


def lvis_forward(pts, view, sd: Dict[str, torch.Tensor]):
    """Lvis.forward (fields.py:361-369): sigmoid(MLP(embed(pts, 10) | embed(view, 4))) -> [M,1]"""
    return torch.sigmoid(sequential_mlp(torch.cat([embed(pts, 10), embed(view, 4)], dim=-1), sd, "lvis"))


def indirect_light_forward(pts, sd: Dict[str, torch.Tensor], num_lgt_sgs: int = 24):
    """IndirectLight.forward (fields.py:399-413): 24 spherical Gaussians per point -> [M,24,7] =
    (lobe axis from two sigmoid angles, sharpness = 30 sigmoid + 0.1, amplitude = relu x 3)"""
    out = sequential_mlp(embed(pts, 10), sd, "indi").reshape(-1, num_lgt_sgs, 6)
    ang = torch.sigmoid(out[..., :2]) * (2.0 * math.pi)
    theta, phi = ang[..., :1], ang[..., 1:2]
    lobes = torch.cat([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], dim=-1)
    lam = torch.sigmoid(out[..., 2:3]) * 30.0 + 0.1
    mu = torch.relu(out[..., 3:])
    return torch.cat([lobes, lam, mu], dim=-1)


def query_indir_illum(lgt_sgs, dirs):
    """calLvis.py:323-336: radiance of [M,L,7] spherical Gaussians towards [M,S,3] directions -> [M,S,3]"""
    lobes = lgt_sgs[:, None, :, :3]
    lobes = lobes / torch.linalg.norm(lobes, dim=-1, keepdim=True)
    lam, mu = lgt_sgs[:, None, :, 3:4], lgt_sgs[:, None, :, 4:]
    cosv = (dirs[:, :, None, :] * lobes).sum(-1, keepdim=True)
    return (mu * torch.exp(lam * (cosv - 1.0))).sum(dim=2)


def sample_dirs(normals, r_theta, r_phi):
    """calLvis.py:302-320: directions at polar angle r_phi from the normal, azimuth r_theta in the tangent frame built
    from the x axis.  normals [M,3] -> [M,S,3]"""
    tiny = 1e-6
    unit = lambda v: v / (torch.linalg.norm(v, dim=-1, keepdim=True) + tiny)
    n = unit(normals)[:, None, :]
    x_axis = torch.zeros_like(n)
    x_axis[..., 0] = 1.0
    U = unit(torch.linalg.cross(x_axis, n, dim=-1))
    V = unit(torch.linalg.cross(n, U, dim=-1))
    th, ph = r_theta[..., None], r_phi[..., None]
    return U * torch.cos(th) * torch.sin(ph) + V * torch.sin(th) * torch.sin(ph) + n * torch.cos(ph)


def first_hit(sdf, mid_z, inside_mask):
    """the first sign change of the SDF along each ray and the depth of the zero crossing by linear interpolation
    (renderer.py:586-602 = calLvis.py:178-194).  sdf, mid_z [R,n]; inside_mask [R] -> (sdf_mask [R] bool, z_surf [R],
    only meaningful where sdf_mask)"""
    R, n = sdf.shape
    ramp = torch.arange(n, 0, -1, dtype=sdf.dtype, device=sdf.device)[None, :]
    val, idx = torch.min(torch.sign(sdf) * ramp, dim=-1)
    mask = (val < 0.0) & (idx >= 1) & inside_mask
    hi = idx.clamp(min=1)[:, None]
    z_lo, z_hi = torch.gather(mid_z, 1, hi - 1), torch.gather(mid_z, 1, hi)
    s_lo, s_hi = torch.gather(sdf, 1, hi - 1), torch.gather(sdf, 1, hi)
    z_surf = (s_lo * z_hi - s_hi * z_lo) / (s_lo - s_hi + 1e-10)
    return mask, z_surf[:, 0]


def secondary_sections(z_vals):
    """calLvis.py:95-100 = :155-160: sections of the 32 fine depths, the last one padded with (1 - 0.1) / 32"""
    sample_dist = (1 - 0.1) / 32.0
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], sample_dist)], -1)
    return dists, z_vals + dists * 0.5


def cal_firHit_rgb(rays_o, rays_d, z_vals, sdf_p, color_p):
    """calLvis.py:153-204: colour of the first surface a secondary ray hits (zeros when it hits nothing)"""
    R, n = z_vals.shape
    _, mid_z = secondary_sections(z_vals)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    sdf = sdf_only(pts, sdf_p).reshape(R, n)
    inside = (torch.linalg.norm(pts, dim=-1).reshape(R, n) < 1.0).to(z_vals.dtype).sum(-1) > 0.0
    mask, z_surf = first_hit(sdf, mid_z, inside)
    hit_rgb = torch.zeros(R, 3, dtype=z_vals.dtype)
    if mask.any():
        p = rays_o[mask] + rays_d[mask] * z_surf[mask][:, None]
        _, feat, normal, _ = sdf_value_feature_normal(p, sdf_p)
        hit_rgb[mask] = color_forward(p, normal, rays_d[mask], feat, color_p)
    return hit_rgb, mask


def compute_weight(rays_o, rays_d, z_vals, sdf_p, inv_s):
    """calLvis.py:93-150: NeuS weights of a secondary ray at cos_anneal_ratio = 0, and the part inside the unit sphere"""
    R, n = z_vals.shape
    dists, mid_z = secondary_sections(z_vals)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    sdf, _, grad, _ = sdf_value_feature_normal(pts, sdf_p)
    dirs = rays_d[:, None, :].expand(R, n, 3).reshape(-1, 3)
    true_cos = (dirs * grad).sum(-1, keepdim=True)
    iter_cos = -torch.relu(-true_cos * 0.5 + 0.5)
    d = dists.reshape(-1, 1)
    prev_cdf = torch.sigmoid((sdf - iter_cos * d * 0.5) * inv_s)
    next_cdf = torch.sigmoid((sdf + iter_cos * d * 0.5) * inv_s)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).reshape(R, n).clip(0.0, 1.0)
    inside = (torch.linalg.norm(pts, dim=-1).reshape(R, n) < 1.0).to(z_vals.dtype)
    weights = alpha * exclusive_transmittance(alpha)
    return weights, weights * inside


def cal_indiLgt(surf, normal, sdf_p, inv_s, color_p, lvis_sd, indi_sd, u_theta, u_z, trace: Optional[dict] = None):
    """calLvis.py:339-409.  u_theta, u_z [M,4]: the two uniform draws of :351-352 (explicit, so that the checker and the
    checked see the same numbers)."""
    M, S = u_theta.shape
    dirs = sample_dirs(normal, u_theta * (2.0 * math.pi), torch.asin(u_z * 0.95))
    o = surf[:, None, :].expand(M, S, 3).reshape(-1, 3)
    d = dirs.reshape(-1, 3)
    with torch.no_grad():
        z_coarse = torch.linspace(0.0, 1.0, 512, dtype=surf.dtype)[None, :].expand(M * S, 512)
        pts = (o[:, None, :] + d[:, None, :] * z_coarse[..., None]).reshape(-1, 3)
        coarse_sdf = sdf_only(pts, sdf_p)
        z_fine = up_sample(o, d, z_coarse, coarse_sdf, 32, inv_s)
        radiance, sec_mask = cal_firHit_rgb(o, d, z_fine, sdf_p, color_p)
        weights, weights_inside = compute_weight(o, d, z_fine, sdf_p, inv_s)
    gt_lvis = (1.0 - weights_inside.sum(-1)).reshape(M, S)
    pre_lvis = lvis_forward(o, d, lvis_sd).reshape(M, S)
    pre_radiance = query_indir_illum(indirect_light_forward(surf, indi_sd), dirs)
    if trace is not None:
        trace.update(dirs=dirs, z_fine=z_fine, sec_sdf_mask=sec_mask, sec_hit_rgb=radiance, sec_weights=weights,
                     sec_weights_inside=weights_inside)
    return {"gt_lvis": gt_lvis, "pre_lvis": pre_lvis, "gt_trace_radiance": radiance.reshape(M, S, 3),
            "pre_trace_radiance": pre_radiance}


def lvis_render(rays_o, rays_d, near, far, sdf_p, inv_s, color_p, lvis_sd, indi_sd, n_samples: int, n_importance: int,
                u_theta, u_z, up_sample_steps: int = 4, trace: Optional[dict] = None):
    """NeuSRenderer.lvis_render (renderer.py:567-627); rows outside sdf_mask keep the value 1"""
    B = rays_o.shape[0]
    with torch.no_grad():
        util = lvis_mateIllu_render_util(rays_o, rays_d, near, far, sdf_p, n_samples, n_importance, up_sample_steps)
        n = util["n_samples"]
        mask, z_surf = first_hit(util["sdf"].reshape(B, n), util["mid_z_vals"], util["inside_sphere_mask"])
    out = {"gt_lvis": torch.ones(B, 4, dtype=rays_o.dtype), "pre_lvis": torch.ones(B, 4, dtype=rays_o.dtype),
           "gt_trace_radiance": torch.ones(B, 4, 3, dtype=rays_o.dtype),
           "pre_trace_radiance": torch.ones(B, 4, 3, dtype=rays_o.dtype), "sdf_mask": mask}
    if mask.any():
        surf = rays_o[mask] + rays_d[mask] * z_surf[mask][:, None]
        with torch.no_grad():
            _, _, normal, _ = sdf_value_feature_normal(surf, sdf_p)
        res = cal_indiLgt(surf, normal, sdf_p, inv_s, color_p, lvis_sd, indi_sd, u_theta, u_z, trace)
        if trace is not None:
            trace.update(normal=normal, pts_surf=surf)
        for k in ("gt_lvis", "pre_lvis", "gt_trace_radiance", "pre_trace_radiance"):
            full = out[k].clone()
            full[mask] = res[k].to(full.dtype)
            out[k] = full
    return out


def stage2_loss(out: dict):
    """lvis.py:164-170: L1 visibility + L1 traced radiance over the rays that hit the surface"""
    m = out["sdf_mask"]
    lvis_loss = (out["gt_lvis"] - out["pre_lvis"]).abs().sum() / (m[:, None].expand(-1, 4).sum() + 1e-6)
    err = (out["gt_trace_radiance"] - out["pre_trace_radiance"]) * m[:, None, None]
    radiance_loss = err.abs().sum() / (m[:, None, None].expand(-1, 4, 3).sum() + 1e-6)
    return {"loss": lvis_loss + radiance_loss, "lvis_loss": lvis_loss, "trace_radiance_loss": radiance_loss}


# ======================================================================================
# Stage 3 (mateIllu.py): material / illumination estimation with spherical Gaussians (models/inverRender.py)
# ======================================================================================
SG_TINY = 1e-6                      # inverRender.py:12


def srgb_to_linear(srgb):
    """math_utils.py:147-152"""
    eps = torch.finfo(torch.float32).eps
    low = 25.0 / 323.0 * srgb
    high = torch.clamp((200.0 * srgb + 11.0) / 211.0, min=eps) ** (12.0 / 5.0)
    return torch.where(srgb <= 0.04045, low, high)


def _unit_tiny(v):
    """inverRender.py:57-58 norm_axis"""
    return v / (torch.linalg.norm(v, dim=-1, keepdim=True) + SG_TINY)


def leaky_mlp(x, sd: Dict[str, torch.Tensor], prefix: str, n_lin: int, slope: float = 0.2):
    """nn.Sequential(Linear, LeakyReLU(0.2), ..., Linear) (inverRender.py:473-507)"""
    for i in range(n_lin):
        x = x @ sd[f"{prefix}.{2 * i}.weight"].t() + sd[f"{prefix}.{2 * i}.bias"]
        if i < n_lin - 1:
            x = F.leaky_relu(x, slope)
    return x


def sg_lambda_trick(lobe1, lam1, mu1, lobe2, lam2, mu2):
    """product of two spherical Gaussians, assuming lam1 << lam2 (inverRender.py:83-103)"""
    ratio = lam1 / (lam2 + SG_TINY)
    lobe1, lobe2 = _unit_tiny(lobe1), _unit_tiny(lobe2)
    dot = (lobe1 * lobe2).sum(-1, keepdim=True)
    t = torch.sqrt(ratio * ratio + 1.0 + 2.0 * ratio * dot + SG_TINY)
    t = torch.minimum(t, ratio + 1.0)
    lam3 = lam2 * t
    lobes = (ratio / (t + SG_TINY)) * lobe1 + (1.0 / (t + SG_TINY)) * lobe2
    mus = mu1 * mu2 * torch.exp(lam2 * (t - ratio - 1.0))
    return lobes, lam3, mus


def sg_hemisphere_int(lam, cos_beta):
    """integral of a spherical Gaussian over the hemisphere around the normal (inverRender.py:106-125)"""
    lam = torch.clamp(lam, min=SG_TINY)
    inv = 1.0 / (lam + SG_TINY)
    t = torch.sqrt(lam + SG_TINY) * (1.6988 + 10.8438 * inv) / (1.0 + 6.2201 * inv + 10.2415 * inv * inv + SG_TINY)
    inv_a = torch.exp(-t)
    up = (cos_beta >= 0).to(lam.dtype)
    inv_b = torch.exp(-t * torch.clamp(cos_beta, min=0.0))
    s1 = (1.0 - inv_a * inv_b) / (1.0 - inv_a + inv_b - inv_a * inv_b + SG_TINY)
    b = torch.exp(t * torch.clamp(cos_beta, max=0.0))
    s2 = (b - inv_a) / ((1.0 - inv_a) * (b + 1.0) + SG_TINY)
    s = up * s1 + (1.0 - up) * s2
    a_b = 2.0 * math.pi / lam * (torch.exp(-lam) - torch.exp(-2.0 * lam))
    a_u = 2.0 * math.pi / lam * (1.0 - torch.exp(-lam))
    return a_b * (1.0 - s) + a_u * s


def sg_integrate_rgb(normal, lobes, lams, mus):
    """sum over the lobes of the hemispherical integral of SG x clamped cosine (inverRender.py:264-283)"""
    mu_cos, lam_cos, alpha_cos = 32.7080, 0.0315, 31.7003
    lobe_p, lam_p, mu_p = sg_lambda_trick(normal, lam_cos, mu_cos, lobes, lams, mus)
    dot1 = torch.clamp((lobe_p * normal).sum(-1, keepdim=True), min=0.0)
    dot2 = torch.clamp((lobes * normal).sum(-1, keepdim=True), min=0.0)
    rgb = mu_p * sg_hemisphere_int(lam_p, dot1) - mus * alpha_cos * sg_hemisphere_int(lams, dot2)
    return torch.clamp(rgb.sum(dim=-2), min=0.0, max=1.0)


def diffuse_visibility(points, normals, lvis_sd, lobes, lams, u_theta, u_phi):
    """per-lobe light visibility from the distilled Lvis network (inverRender.py:128-192).  lobes [M,3], lams [M,1] of the
    direct-light SGs; u_theta, u_phi [M,S]: the uniform draws of :152-153.  -> [M, n]"""
    M, S = u_theta.shape
    n = points.shape[0]
    axis = _unit_tiny(lobes.detach()[:, None, :])                         # [M,1,3]
    z = torch.zeros_like(axis)
    z[..., 2] = 1.0
    U = _unit_tiny(torch.linalg.cross(z, axis, dim=-1))
    V = _unit_tiny(torch.linalg.cross(axis, U, dim=-1))
    sharp = lams.detach()[:, None, :][:, :, 0]                            # [M,1]
    phi_range = torch.arccos((-1.95 * sharp.min()) / sharp + 1.0)
    th = (u_theta * 2 * math.pi)[..., None]
    ph = (u_phi * phi_range)[..., None]
    dirs = U * torch.cos(th) * torch.sin(ph) + V * torch.sin(th) * torch.sin(ph) + axis * torch.cos(ph)      # [M,S,3]
    flat = dirs.reshape(-1, 3)
    d_all = flat[None, :, :].expand(n, -1, 3)
    p_all = points[:, None, :].expand(-1, M * S, 3)
    front = (normals[:, None, :] * d_all).sum(-1) > SG_TINY
    vis = torch.zeros(n, M * S, dtype=points.dtype)
    with torch.no_grad():
        vis[front] = lvis_forward(p_all[front], d_all[front], lvis_sd).reshape(-1)
    vis = vis.reshape(n, M, S).permute(1, 2, 0)                           # [M,S,n]
    w = torch.exp(lams.detach()[:, None, :] * ((dirs * axis).sum(-1, keepdim=True) - 1.0))       # [M,S,1]
    return ((vis * w).sum(dim=1) / (w.sum(dim=1) + SG_TINY)).detach()


def render_with_sg(points, normal, viewdirs, lgtSGs, specular_reflectance, specular_albedo, roughness, diffuse_albedo,
                   light_vis=None):
    """inverRender.py:314-449.  lgtSGs [n,M,7]; light_vis [M,n] or None (comp_vis=False)
    -> linear env_rgb, tone-mapped diffuse / specular, mean visibility"""
    n, M = lgtSGs.shape[0], lgtSGs.shape[1]
    lobes = lgtSGs[..., :3] / (torch.linalg.norm(lgtSGs[..., :3], dim=-1, keepdim=True) + SG_TINY)
    lams = torch.abs(lgtSGs[..., 3:4])
    mus0 = torch.abs(lgtSGs[..., -3:])
    nrm = normal[:, None, :].expand(n, M, 3)
    view = viewdirs[:, None, :].expand(n, M, 3).detach()
    # normal-distribution function as an SG around the normal, warped to the reflected direction
    inv_r4 = 2.0 / (roughness * roughness * roughness * roughness)        # [n,1]
    brdf_lam = inv_r4[:, None, :].expand(n, M, 1)
    brdf_mu = (inv_r4 / math.pi).expand(n, 3)[:, None, :].expand(n, M, 3)
    v_dot_lobe = torch.clamp((nrm * view).sum(-1, keepdim=True), min=0.0)
    warp_lobes = 2 * v_dot_lobe * nrm - view
    warp_lobes = warp_lobes / (torch.linalg.norm(warp_lobes, dim=-1, keepdim=True) + SG_TINY)
    warp_lams = brdf_lam / (4 * v_dot_lobe + SG_TINY)
    half = warp_lobes + view
    half = half / (torch.linalg.norm(half, dim=-1, keepdim=True) + SG_TINY)
    v_dot_h = torch.clamp((view * half).sum(-1, keepdim=True), min=0.0)
    f0 = specular_reflectance[:, None, :].expand(n, M, 3)
    fresnel = f0 + (1.0 - f0) * torch.pow(2.0, -(5.55473 * v_dot_h + 6.8316) * v_dot_h)
    dot1 = torch.clamp((warp_lobes * nrm).sum(-1, keepdim=True), min=0.0)
    dot2 = torch.clamp((view * nrm).sum(-1, keepdim=True), min=0.0)
    k = ((roughness + 1.0) * (roughness + 1.0) / 8.0)[:, None, :].expand(n, M, 1)
    g1 = dot1 / (dot1 * (1 - k) + k + SG_TINY)
    g2 = dot2 / (dot2 * (1 - k) + k + SG_TINY)
    moi = fresnel * (g1 * g2) / (4 * dot1 * dot2 + SG_TINY)
    warp_mus = specular_albedo[:, None, :] * brdf_mu * moi
    shadow = torch.zeros(n, 3, dtype=points.dtype)
    if light_vis is not None:
        vis = light_vis.permute(1, 0)[..., None].expand(n, M, 3)
        mus = mus0 * vis
        shadow = vis.mean(dim=1)
    else:
        mus = mus0
    # specular: (light SG x BRDF SG) x cosine
    fl, fla, fmu = sg_lambda_trick(lobes, lams, mus, warp_lobes, warp_lams, warp_mus)
    specular = sg_integrate_rgb(nrm, fl, fla, fmu)
    # diffuse: light SG x albedo / pi x cosine
    diffuse = sg_integrate_rgb(nrm, lobes, lams, mus * (diffuse_albedo / math.pi)[:, None, :].expand(n, M, 3))
    return {"env_rgb": torch.clamp(specular + diffuse, 0.0, 1.0),
            "diffuse_rgb": torch.clip(linear_to_srgb(diffuse), 0.0, 1.0),
            "specular_rgb": torch.clip(linear_to_srgb(specular), 0.0, 1.0), "lvis_mean": shadow}


def envmap_material_forward(points, ray_dirs, n, indi_lgt, lvis_sd, mat_sd, u_theta, u_phi, specular_reflectance: float = 0.02):
    """EnvmapMaterialNetwork.forward (inverRender.py:530-598) + render_with_all_sg (:286-311)"""
    n = n / (torch.linalg.norm(n, dim=-1, keepdim=True) + SG_TINY)
    ray_dirs = ray_dirs / (torch.linalg.norm(ray_dirs, dim=-1, keepdim=True) + SG_TINY)
    view = -ray_dirs
    pts_enc = embed(points, 10)
    latent_pre = leaky_mlp(pts_enc, mat_sd, "brdf_encoder_layer", 5)
    brdf = torch.sigmoid(leaky_mlp(torch.sigmoid(latent_pre), mat_sd, "brdf_decoder_layer", 3))
    roughness = brdf[..., 3:] * 0.9 + 0.09
    diffuse_albedo = brdf[..., :3]
    # sparsity of the latent code: KL(0.05 || mean sigmoid)  (inverRender.py:553-559, 609-612)
    rho, rho_hat = 0.05, torch.sigmoid(latent_pre).mean(0)
    kl = (rho * torch.log(rho / rho_hat) + (1 - rho) * torch.log((1 - rho) / (1 - rho_hat))).mean()
    spec_in = torch.cat([pts_enc, embed(reflect(view, n), 4)], dim=-1)
    specular_albedo = torch.sigmoid(leaky_mlp(spec_in, mat_sd, "net_cs", 5)).repeat(1, 3)
    lgt = mat_sd["lgtSGs"]
    f0 = torch.full((1, 1), specular_reflectance, dtype=points.dtype)
    direct_sgs = lgt[None].expand(points.shape[0], -1, -1)
    lobes0 = direct_sgs[0, :, :3] / (torch.linalg.norm(direct_sgs[0, :, :3], dim=-1, keepdim=True) + SG_TINY)
    vis = diffuse_visibility(points, n, lvis_sd, lobes0, torch.abs(direct_sgs[0, :, 3:4]), u_theta, u_phi)
    ret = render_with_sg(points, n, view, direct_sgs, f0, specular_albedo, roughness, diffuse_albedo, light_vis=vis)
    indir = render_with_sg(points, n, view, indi_lgt, f0, specular_albedo, roughness, diffuse_albedo)["env_rgb"]
    env = ret["env_rgb"]
    ret.update(rgb=torch.clip(linear_to_srgb(env + indir), 0.0, 1.0), indir_rgb=torch.clip(linear_to_srgb(indir), 0.0, 1.0),
               env_rgb=torch.clip(linear_to_srgb(env), 0.0, 1.0), roughness=roughness,
               diffuse_albedo=torch.clip(linear_to_srgb(diffuse_albedo), 0.0, 1.0),
               specular_albedo=torch.clip(linear_to_srgb(specular_albedo), 0.0, 1.0), encoder_loss=0.01 * kl, light_vis=vis)
    return ret


def mateIllu_render(rays_o, rays_d, near, far, sdf_p, refcolor_sd, lvis_sd, indi_sd, mat_sd, n_samples: int, n_importance: int,
                    u_theta, u_phi, up_sample_steps: int = 4):
    """NeuSRenderer.mateIllu_render (renderer.py:630-726); rows outside sdf_mask keep the value 1"""
    B = rays_o.shape[0]
    with torch.no_grad():
        util = lvis_mateIllu_render_util(rays_o, rays_d, near, far, sdf_p, n_samples, n_importance, up_sample_steps)
        n = util["n_samples"]
        mask, z_surf = first_hit(util["sdf"].reshape(B, n), util["mid_z_vals"], util["inside_sphere_mask"])
    one3 = lambda: torch.ones(B, 3, dtype=rays_o.dtype)
    out = {k: one3() for k in ("rgb", "env_rgb", "indir_rgb", "diffuse_albedo", "specular_albedo", "diffuse_rgb", "specular_rgb",
                               "lvis_mean", "n_out", "gt_specular_linear", "gt_diffuse_srgb")}
    out["roughness"] = torch.ones(B, 1, dtype=rays_o.dtype)
    out.update(sdf_mask=mask, encoder_loss=torch.zeros(()), diffuse_loss=0, specular_loss=0, smooth_loss=0)
    if mask.any():
        pts = rays_o[mask] + rays_d[mask] * z_surf[mask][:, None]
        with torch.no_grad():
            _, feat, n_surf, _ = sdf_value_feature_normal(pts, sdf_p)
            rc = refcolor_forward(pts, feat, rays_d[mask], n_surf, refcolor_sd)
            spec_lin = srgb_to_linear(rc["specular_rgb"])
            indi = indirect_light_forward(pts, indi_sd)
        m = envmap_material_forward(pts, rays_d[mask], n_surf, indi, lvis_sd, mat_sd, u_theta, u_phi)
        for k in ("rgb", "env_rgb", "indir_rgb", "diffuse_albedo", "specular_albedo", "diffuse_rgb", "specular_rgb", "roughness",
                  "lvis_mean"):
            full = out[k].clone()
            full[mask] = m[k].to(full.dtype)
            out[k] = full
        for k, v in (("gt_specular_linear", spec_lin), ("gt_diffuse_srgb", rc["diffuse_rgb"]), ("n_out", n_surf)):
            full = out[k].clone()
            full[mask] = v
            out[k] = full
        out["encoder_loss"] = m["encoder_loss"]
        out["_light_vis"] = m["light_vis"]
    return out


def stage3_loss(out: dict, true_rgb, mask):
    """mateIllu.py:152-172: masked L1 colour over the rays that hit + the latent sparsity term"""
    m = out["sdf_mask"]
    denom = mask[m].sum() + 1e-5
    rgb_loss = ((out["rgb"][m] - true_rgb[m]) * mask[m]).abs().sum() / denom
    psnr = 20.0 * torch.log10(1.0 / (((out["rgb"][m] - true_rgb[m]) ** 2 * mask[m]).sum() / (denom * 3.0)).sqrt())
    return {"loss": rgb_loss + out["encoder_loss"], "rgb_loss": rgb_loss, "encoder_loss": out["encoder_loss"], "psnr": psnr}
