"""Deterministic synthetic inputs for the hot path: network weights and DTU-shaped ray batches.

Used by bench.py, the tests and tests/golden/gen_golden.py.  numpy RandomState streams only, so the
same seed gives bit-identical arrays in the build container and on the GPU box (no torch RNG).

Weight distributions follow the reference initialisers:
  SDFNetwork geometric init ........ models/fields.py:47-65 (then weight_norm, :67-68)
  RenderingNetwork / NeRF / RefColor  default nn.Linear init (kaiming-uniform(a=sqrt 5) = U(+-1/sqrt(fan_in)))
  SingleVarianceNetwork ............ init_val 0.3 (confs/wmask.conf:75)
plus a small deterministic perturbation so the SDF is not an exact sphere and g != ||v||.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

SDF_DIMS = [39, 256, 256, 256, 256, 256, 256, 256, 256, 257]   # fields.py:26-34 with multires=6
SDF_SKIP = (4,)
COLOR_DIMS = [289, 256, 256, 256, 256, 3]                       # fields.py:127-133 (9+256+24)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sdf_state_dict(seed: int = 0, perturb: float = 0.02, bias: float = 0.5, warp=None, inside_out: bool = False) -> "OrderedDict[str, np.ndarray]":
    """lin{l}.bias / weight_g / weight_v in the reference's state_dict order (fields.py:67-70).
    warp = (k, c): the first layer also sees c * sin(2^k x) beside x, i.e. the sphere of the geometric initialisation in
    the coordinates x + c sin(2^k x) -- a bumpy surface.  inside_out: the sign convention of geometric_init with
    inside_outside=True (fields.py:50-52): positive INSIDE the sphere -- a room seen from within, so that the secondary rays
    of stage 2 (calLvis.py:339-409) leave a wall and hit the opposite one."""
    rs = np.random.RandomState(seed)
    dims = SDF_DIMS
    n_lin = len(dims) - 1
    sd = OrderedDict()
    for l in range(n_lin):
        out_dim = dims[l + 1] - dims[0] if (l + 1) in SDF_SKIP else dims[l + 1]
        in_dim = dims[l]
        if l == n_lin - 1:
            w = rs.normal(math.sqrt(math.pi) / math.sqrt(in_dim), 1e-4, size=(out_dim, in_dim))
            b = np.full((out_dim,), -bias)
            if inside_out:
                w[0], b[0] = -w[0], -b[0]
        elif l == 0:
            w = np.zeros((out_dim, in_dim))
            w[:, :3] = rs.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, 3))
            if warp is not None:                   # embedding columns: x, then (sin 2^k x, cos 2^k x) per octave (embedder.py:35-36)
                k, c = warp
                w[:, 3 + 6 * k: 6 + 6 * k] = c * w[:, :3]
            b = np.zeros((out_dim,))
        elif l in SDF_SKIP:
            w = rs.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, in_dim))
            w[:, -(dims[0] - 3):] = 0.0
            b = np.zeros((out_dim,))
        else:
            w = rs.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, in_dim))
            b = np.zeros((out_dim,))
        if perturb > 0:
            scale = np.abs(w).mean() + 1e-3
            w = w + perturb * scale * rs.standard_normal(w.shape)
            b = b + perturb * 0.05 * rs.standard_normal(b.shape)
        g = np.linalg.norm(w, axis=1, keepdims=True)
        if perturb > 0:
            g = g * (1.0 + perturb * rs.standard_normal(g.shape))
        sd[f"lin{l}.bias"] = _f32(b)
        sd[f"lin{l}.weight_g"] = _f32(g)
        sd[f"lin{l}.weight_v"] = _f32(w)
    return sd


def _linear_default(rs, out_dim, in_dim):
    k = 1.0 / math.sqrt(in_dim)
    return rs.uniform(-k, k, size=(out_dim, in_dim)), rs.uniform(-k, k, size=(out_dim,))


def color_state_dict(seed: int = 1) -> "OrderedDict[str, np.ndarray]":
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for l in range(len(COLOR_DIMS) - 1):
        w, b = _linear_default(rs, COLOR_DIMS[l + 1], COLOR_DIMS[l])
        g = np.linalg.norm(w, axis=1, keepdims=True) * (1.0 + 0.02 * rs.standard_normal((w.shape[0], 1)))
        sd[f"lin{l}.bias"] = _f32(b)
        sd[f"lin{l}.weight_g"] = _f32(g)
        sd[f"lin{l}.weight_v"] = _f32(w)
    return sd


def refcolor_state_dict(seed: int = 2) -> "OrderedDict[str, np.ndarray]":
    """RefColor (fields.py:271-300): net_cd.{0,2,4,6,8}, viewdir_mlp.{0..3}, net_cs.0."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    dims_cd = [286, 256, 256, 256, 256, 3]
    for i in range(5):
        w, b = _linear_default(rs, dims_cd[i + 1], dims_cd[i])
        sd[f"net_cd.{2 * i}.weight"], sd[f"net_cd.{2 * i}.bias"] = _f32(w), _f32(b)
    dims_vm = [289, 256, 256, 256, 256]
    for i in range(4):
        # NB the reference builds four LazyLinear(256); layers 1..3 therefore take 256 inputs (fields.py:295-297)
        w, b = _linear_default(rs, dims_vm[i + 1], dims_vm[i])
        sd[f"viewdir_mlp.{i}.weight"], sd[f"viewdir_mlp.{i}.bias"] = _f32(w), _f32(b)
    w, b = _linear_default(rs, 1, 256)
    sd["net_cs.0.weight"], sd["net_cs.0.bias"] = _f32(w), _f32(b)
    return sd


LVIS_DIMS = [90, 256, 256, 256, 256, 1]          # fields.py:338-369: embed(pts, 10) | embed(view, 4) -> 256 x 4 -> 1
INDI_DIMS = [63, 512, 512, 512, 512, 144]         # fields.py:372-413: embed(pts, 10) -> 512 x 4 -> 24 x 6


def _sequential_state_dict(rs, prefix, dims, last_gain=1.0):
    sd = OrderedDict()
    for i in range(len(dims) - 1):
        w, b = _linear_default(rs, dims[i + 1], dims[i])
        if i == len(dims) - 2:
            w, b = w * last_gain, b * last_gain
        sd[f"{prefix}.{2 * i}.weight"], sd[f"{prefix}.{2 * i}.bias"] = _f32(w), _f32(b)
    return sd


def lvis_state_dict(seed: int = 4) -> "OrderedDict[str, np.ndarray]":
    """Lvis (fields.py:338-369): lvis.{0,2,4,6,8}; the last layer is scaled up so that the sigmoid leaves 0.5"""
    return _sequential_state_dict(np.random.RandomState(seed), "lvis", LVIS_DIMS, last_gain=8.0)


def indilgt_state_dict(seed: int = 5) -> "OrderedDict[str, np.ndarray]":
    """IndirectLight (fields.py:372-413): indi.{0,2,4,6,8}; the last layer is scaled up so that lobes, sharpness and
    amplitudes spread over their ranges"""
    return _sequential_state_dict(np.random.RandomState(seed), "indi", INDI_DIMS, last_gain=6.0)


def fibonacci_sphere(samples: int) -> np.ndarray:
    """inverRender.py:69-82: golden-angle spiral, y from 1 to -1"""
    i = np.arange(samples, dtype=np.float64)
    y = 1.0 - (i / float(samples - 1)) * 2.0
    radius = np.sqrt(1.0 - y * y)
    theta = np.pi * (3.0 - np.sqrt(5.0)) * i
    return np.stack([np.cos(theta) * radius, y, np.sin(theta) * radius], axis=1)


def mateillu_state_dict(seed: int = 6, num_lgt_sgs: int = 128) -> "OrderedDict[str, np.ndarray]":
    """EnvmapMaterialNetwork (inverRender.py:451-528): lgtSGs [128,7] following the reference's initialisation recipe
    (grey amplitudes, sharpness 10 + 20 |N|, envmap energy normalised to 0.8 * 2 pi, lobes on two Fibonacci spheres), then
    brdf_encoder_layer.{0,..,8} (63 -> 512 x 4 -> 32), brdf_decoder_layer.{0,2,4} (32 -> 128 x 2 -> 4), net_cs.{0,..,8}
    (90 -> 256 x 4 -> 1) in the module's state_dict order."""
    rs = np.random.RandomState(seed)
    sg = rs.standard_normal((num_lgt_sgs, 7))
    sg[:, -2:] = sg[:, -3:-2]
    sg[:, 3:4] = 10.0 + np.abs(sg[:, 3:4] * 20.0)
    lam, mu = np.abs(sg[:, 3:4]), np.abs(sg[:, 4:])
    energy = mu * 2.0 * np.pi / lam * (1.0 - np.exp(-2.0 * lam))
    sg[:, 4:] = np.abs(sg[:, 4:]) / energy.sum(axis=0, keepdims=True) * 2.0 * np.pi * 0.8
    lobes = fibonacci_sphere(num_lgt_sgs // 2)
    sg[: num_lgt_sgs // 2, :3] = lobes
    sg[num_lgt_sgs // 2:, :3] = lobes
    sd = OrderedDict()
    sd["lgtSGs"] = _f32(sg)
    sd.update(_sequential_state_dict(rs, "brdf_encoder_layer", [63, 512, 512, 512, 512, 32], last_gain=3.0))
    sd.update(_sequential_state_dict(rs, "brdf_decoder_layer", [32, 128, 128, 4], last_gain=6.0))
    sd.update(_sequential_state_dict(rs, "net_cs", [90, 256, 256, 256, 256, 1], last_gain=6.0))
    return sd


def nerf_state_dict(seed: int = 3, D=8, W=256, input_ch=84, input_ch_view=27) -> "OrderedDict[str, np.ndarray]":
    """NeRF (fields.py:178-231) with use_viewdirs=True, skips=[4]."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for i in range(D):
        in_dim = input_ch if i == 0 else (W + input_ch if (i - 1) == 4 else W)
        w, b = _linear_default(rs, W, in_dim)
        sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"] = _f32(w), _f32(b)
    w, b = _linear_default(rs, W // 2, input_ch_view + W)
    sd["views_linears.0.weight"], sd["views_linears.0.bias"] = _f32(w), _f32(b)
    w, b = _linear_default(rs, W, W)
    sd["feature_linear.weight"], sd["feature_linear.bias"] = _f32(w), _f32(b)
    w, b = _linear_default(rs, 1, W)
    sd["alpha_linear.weight"], sd["alpha_linear.bias"] = _f32(w), _f32(b)
    w, b = _linear_default(rs, 3, W // 2)
    sd["rgb_linear.weight"], sd["rgb_linear.bias"] = _f32(w), _f32(b)
    return sd


def ray_batch(batch: int, seed: int = 0, radius: float = 2.8, half_extent: float = 0.4,
              mask_prob: float = 0.7, n_miss: int = 0) -> np.ndarray:
    """DTU-shaped batch [B,10] = rays_o(3), rays_d(3), rgb(3), mask(1) from ONE camera (dataset.py:133-151).

    Camera centre uniform on the sphere of `radius`; pixel targets uniform in [-h,h]^3; the last `n_miss`
    rays are aimed away from the unit sphere (they never enter it).
    """
    rs = np.random.RandomState(seed)
    c = rs.standard_normal(3)
    c = c / np.linalg.norm(c) * radius
    tgt = rs.uniform(-half_extent, half_extent, size=(batch, 3))
    if n_miss > 0:
        tgt[-n_miss:] = c[None, :] * 0.5 + rs.uniform(1.5, 2.5, size=(n_miss, 3)) * np.sign(rs.standard_normal((n_miss, 3)))
    d = tgt - c[None, :]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    rgb = rs.uniform(0, 1, size=(batch, 3))
    mask = (rs.uniform(0, 1, size=(batch, 1)) < mask_prob).astype(np.float64)
    o = np.broadcast_to(c[None, :], (batch, 3))
    return _f32(np.concatenate([o, d, rgb, mask], axis=1))


def icosphere(level: int = 3):
    """unit icosphere: 12 vertices subdivided `level` times -> (vertices [V,3] float64, triangles [T,3] int64); deterministic"""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        verts = list(map(tuple, v))
        cache, faces = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (np.asarray(verts[a]) + np.asarray(verts[b])) * 0.5
                verts.append(tuple(m / np.linalg.norm(m)))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            faces += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)
    return v, f


def dtu_eval_scene(seed: int = 0):
    """A DTU-shaped evaluation case in millimetres (evaluation/dtu_eval.py:36-162): a reconstructed mesh (icosphere of
    radius 12 with a smooth bump), the "scanned" reference cloud (points on the true sphere of radius 11.8, above and below
    the ground plane), an observability grid with one octant masked out and the ground plane."""
    rs = np.random.RandomState(seed)
    v, f = icosphere(3)
    bump = 1.0 + 0.03 * np.sin(3.0 * v[:, 0]) * np.cos(2.0 * v[:, 1])
    vertices = v * (12.0 * bump)[:, None] + np.array([5.0, -3.0, 20.0])
    n = 30000
    g = rs.standard_normal((n, 3))
    stl = 11.8 * g / np.linalg.norm(g, axis=1, keepdims=True) + np.array([5.0, -3.0, 20.0])
    bb = np.array([[-12.0, -20.0, 4.0], [22.0, 14.0, 36.0]])
    res = 2.0
    shape = tuple(int(round((bb[1, i] - bb[0, i]) / res)) + 1 for i in range(3))
    obs = np.ones(shape, dtype=np.uint8)
    obs[: shape[0] // 2, : shape[1] // 2, : shape[2] // 2] = 0            # one octant of the volume was never observed
    plane = np.array([0.0, 0.0, 1.0, -12.0])                               # points with z > 12 count for stl -> data
    return {"vertices": vertices, "triangles": f, "stl": stl, "ObsMask": obs, "BB": bb, "Res": res, "P": plane}
