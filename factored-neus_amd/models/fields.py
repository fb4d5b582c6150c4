"""Network modules with the reference's class API and state_dict keys (models/fields.py), HIP backend.

  SDFNetwork / RenderingNetwork : parameters live here as torch Parameters named exactly like the reference
      (lin{l}.weight_g / weight_v / bias: old-style nn.utils.weight_norm, fields.py:67-70, 139-140) so checkpoints
      interchange; the maths runs in libfneus_hip.so.  Only the architecture of confs/wmask.conf / womask.conf is
      supported by the fused kernels -- anything else raises (there is no fallback path).
  RefColor : the surface head; its two MLPs run on the colour-network kernels (plain Linear layers).
  NeRF : the womask background NeRF++, plain Linear layers on the fused K7 kernels.
  SingleVarianceNetwork : a scalar.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from fneus import ops
from fneus.autograd import RaySamples, SdfValueGradFn, ColorFn, RefHeadsFn, NerfFn, _Workspace
from models.embedder import get_embedder


class WNLinear(nn.Module):
    """Parameters of a weight-normalised Linear with the reference's names: weight_g [out,1], weight_v [out,in], bias."""

    def __init__(self, in_dim, out_dim, weight_norm=True):
        super().__init__()
        self.weight_norm = weight_norm
        w = torch.empty(out_dim, in_dim)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_dim)
        b = torch.empty(out_dim).uniform_(-bound, bound)
        self.bias = nn.Parameter(b)
        if weight_norm:
            self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)

    def set_weight(self, w, b):
        with torch.no_grad():
            self.bias.copy_(b)
            if self.weight_norm:
                self.weight_v.copy_(w)
                self.weight_g.copy_(w.norm(dim=1, keepdim=True))
            else:
                self.weight.copy_(w)

    def effective_weight(self):
        if not self.weight_norm:
            return self.weight
        return self.weight_v * (self.weight_g / self.weight_v.norm(dim=1, keepdim=True))


def _frozen_key(params):
    """(address, version) of every parameter when NONE of them is trained (the frozen networks of stages 2 / 3), else None: a
    frozen network whose key has not changed since it was packed need not be packed again (a state_dict load or any other in-place
    write bumps the versions; trained parameters are also written by the optimiser kernel, which does not, so they always pack)"""
    params = list(params)
    if any(p.requires_grad for p in params):
        return None
    return tuple((p.data_ptr(), p._version) for p in params)


class _HipMLP(nn.Module):
    """Shared plumbing of the two fused MLPs.

    The nn.Parameters (reference names lin{l}.bias / weight_g / weight_v) are VIEWS into one flat fp32 buffer owned by
    the PackedNet, and their .grad are views into one flat gradient buffer: the packer reads the flat buffer directly
    (weight-norm fold inside the kernel) and the backward kernels accumulate into the flat gradient, so the module
    costs a handful of launches per step; the data-parallel all-reduce and Adam see ordinary Parameters."""

    kind = None

    def _init_backend(self):
        self._net = None
        self._ws = _Workspace()
        self._anchor = None
        self._packed = False
        self._ext_grad = None          # optional slice of a model-wide gradient arena (set before the first refresh)
        self.prec = ops.PREC_PARITY

    def set_precision(self, prec: int):
        assert prec in (ops.PREC_FAST, ops.PREC_PARITY)
        self.prec = prec

    def set_gradient_precision(self, gprec):
        """1: the backward stash holds bf16 planes (default: the weight-gradient products carry 2^-9 rounding each), 3: hi +
        lo planes (fp32-accurate weight gradients, twice the stash traffic), 2: bf16 planes but for the two operands of the colour
        network's output layer -- the one product whose rounding exceeds mode 3's bounds (ops._gprec); None: ops.DEFAULT_GPREC.
        Forward outputs do not depend on it."""
        assert gprec in (None, 1, 2, 3)
        self._ws.gprec = gprec
        for k in [k for k in self._ws.cache if k[0] in ("sdf_stash", "sdf_bwd", "sdf_jobs", "col_stash", "col_jobs", "col_out_jobs")]:
            del self._ws.cache[k]

    def _lins(self):
        return [getattr(self, f"lin{l}") for l in range(self.num_layers - 1)]

    def _attach(self):
        """(re)create the device backend and alias the Parameters onto its flat buffers"""
        dev = self.lin0.bias.device
        if dev.type != "cuda":
            raise RuntimeError("the fneus HIP backend needs the module on a GPU (there is no CPU fallback)")
        if any(not lin.weight_norm for lin in self._lins()):
            raise NotImplementedError("the fused packer expects weight_norm=True as in confs/wmask.conf")
        net = ops.PackedNet(self.kind, dev, raw_grad=self._ext_grad)
        with torch.no_grad():
            for lin, view, gview in zip(self._lins(), net.raw_views(net.raw), net.raw_views(net.raw_grad)):
                for name in ("bias", "weight_g", "weight_v"):
                    prm = getattr(lin, name)
                    view[name].copy_(prm.data)
                    if prm.grad is not None:
                        gview[name].copy_(prm.grad)
                    prm.data = view[name]
                    prm.grad = gview[name]
        self._net = net
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)

    def n_raw(self) -> int:
        """number of raw parameter values (= size of the flat gradient buffer)"""
        return sum(p.numel() for p in self.parameters())

    def use_grad_buffer(self, buf: torch.Tensor):
        """put the flat gradient buffer into `buf` (a slice of a model-wide arena); call before the first refresh()"""
        self._ext_grad = buf
        self._net = None
        self._ws.cache.clear()         # job tables cache gradient-buffer addresses

    def _attached(self):
        return (self._net is not None and self.lin0.bias.device == self._net.device
                and self.lin0.bias.data_ptr() == self._net.raw_views(self._net.raw)[0]["bias"].data_ptr())

    def refresh(self):
        """Fold weight-norm and pack the current parameters for the kernels.  Call once per step before rendering."""
        if not self._attached():
            self._attach()
            self._frozen_packed = None
        key = _frozen_key(self.parameters())
        if key is not None and key == getattr(self, "_frozen_packed", None) and self._packed:
            return
        self._frozen_packed = key
        for lin, gview in zip(self._lins(), self._net.raw_views(self._net.raw_grad)):
            for name in ("bias", "weight_g", "weight_v"):      # optimizer.zero_grad(set_to_none=True) drops the views
                prm = getattr(lin, name)
                if prm.grad is None:
                    gview[name].zero_()
                    prm.grad = gview[name]
        self._net.pack()
        self._packed = True

    def _ensure(self):
        if not self._packed or not self._attached():
            self.refresh()


class SDFNetwork(_HipMLP):
    kind = "sdf"

    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1, geometric_init=True,
                 weight_norm=True, inside_outside=False):
        super().__init__()
        if not (d_in == 3 and d_out == 257 and d_hidden == 256 and n_layers == 8 and tuple(skip_in) == (4,)
                and multires == 6 and float(scale) == 1.0):
            raise NotImplementedError("fneus HIP kernels are specialised for the reference SDF architecture "
                                      "(confs/wmask.conf:60-71: 3->PE6->8x256 skip 4 ->257, scale 1)")
        dims = [39] + [d_hidden] * n_layers + [d_out]
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = WNLinear(dims[l], out_dim, weight_norm)
            if geometric_init:      # fields.py:47-65
                w = torch.empty(out_dim, dims[l])
                b = torch.zeros(out_dim)
                if l == self.num_layers - 2:
                    sgn = -1.0 if inside_outside else 1.0
                    nn.init.normal_(w, mean=sgn * math.sqrt(math.pi) / math.sqrt(dims[l]), std=0.0001)
                    b.fill_(-sgn * bias)
                elif l == 0:
                    w.zero_()
                    nn.init.normal_(w[:, :3], 0.0, math.sqrt(2) / math.sqrt(out_dim))
                elif l in self.skip_in:
                    nn.init.normal_(w, 0.0, math.sqrt(2) / math.sqrt(out_dim))
                    w[:, -(dims[0] - 3):] = 0.0
                else:
                    nn.init.normal_(w, 0.0, math.sqrt(2) / math.sqrt(out_dim))
                lin.set_weight(w, b)
            setattr(self, "lin" + str(l), lin)
        self._init_backend()

    # ---- hot-path entry points used by NeuSRenderer ----
    def sdf_samples(self, samples: RaySamples, ray_mask=None) -> torch.Tensor:
        """no-grad SDF values [n] (K1).  ray_mask [rays]: rays whose values the caller reads (the others get 1.0 unevaluated)"""
        self._ensure()
        return ops.sdf_fwd(self._net.blob, samples.n, self.prec, ray_mask=ray_mask, **samples.kw())

    def sdf_merge_upsample(self, rays_o, rays_d, z_old, s_old, z_new, inv_s: float, k_next: int, last: bool, sample_dist=None):
        """one step of the hierarchical sampler in one launch: no-grad SDF values at the new depths + cat_z_vals + the next up_sample
        (ops.sdf_merge_upsample; None when the launch does not take the shape)"""
        self._ensure()
        return ops.sdf_merge_upsample(self._net.blob, self.prec, rays_o, rays_d, z_old, s_old, z_new, inv_s, k_next, last, sample_dist)

    def sdf_merge_upsample_steps(self, rays_o, rays_d, z_old, s_old, z_new, inv_s_list, k_next: int, sample_dist: float):
        """every remaining step of the hierarchical sampler in one launch (ops.sdf_merge_upsample_steps; None: not this shape)"""
        self._ensure()
        return ops.sdf_merge_upsample_steps(self._net.blob, self.prec, rays_o, rays_d, z_old, s_old, z_new, inv_s_list, k_next, sample_dist)

    def value_feature_normal(self, samples: RaySamples, train: bool, feat_rows: bool = True):
        """sdf [n], feature [n,256], normal [n,3] in one fused pass (K2), differentiable w.r.t. the parameters.
        feat_rows False: the caller hands `feature` to RenderingNetwork.color_samples / SurfaceGatherFn only -- they read the stash's
        hi + lo feature planes, and a chip-filling training launch then leaves the fp32 rows out (the tensor is a placeholder)"""
        self._ensure()
        return SdfValueGradFn.apply(self._anchor, self._net, samples, self.prec, self._ws, train, feat_rows)

    # ---- reference API (fields.py:74-111) ----
    def forward(self, inputs, iter_step=0):
        pts = inputs.detach().reshape(-1, 3).float().contiguous()
        sdf, feat, _ = self.value_feature_normal(RaySamples(pts=pts), train=torch.is_grad_enabled())
        return torch.cat([sdf[:, None], feat], dim=-1)

    def sdf(self, x):
        pts = x.detach().reshape(-1, 3).float().contiguous()
        return self.sdf_samples(RaySamples(pts=pts))[:, None]

    def sdf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        pts = x.detach().reshape(-1, 3).float().contiguous()
        _, _, normal = self.value_feature_normal(RaySamples(pts=pts), train=torch.is_grad_enabled())
        return normal.unsqueeze(1)


class RenderingNetwork(_HipMLP):
    kind = "color"

    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=0,
                 squeeze_out=True):
        super().__init__()
        if not (d_feature == 256 and mode == "idr" and d_in == 9 and d_out == 3 and d_hidden == 256 and n_layers == 4
                and multires_view == 4 and squeeze_out):
            raise NotImplementedError("fneus HIP kernels are specialised for the reference colour architecture "
                                      "(confs/wmask.conf:77-87: idr, 289->4x256->3, PE4 view, sigmoid)")
        self.mode, self.squeeze_out = mode, squeeze_out
        dims = [d_in + d_feature + 24] + [d_hidden] * n_layers + [d_out]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            setattr(self, "lin" + str(l), WNLinear(dims[l], dims[l + 1], weight_norm))
        self._init_backend()

    def color_samples(self, samples: RaySamples, normal, feat, sdf_net: SDFNetwork, train: bool):
        self._ensure()
        return ColorFn.apply(self._anchor, normal, feat, self._net, samples, self.prec, self._ws, sdf_net._ws, train)

    def forward(self, points, normals, view_dirs, feature_vectors):
        s = RaySamples(pts=points.detach().float().contiguous(), dirs=view_dirs.detach().float().contiguous())
        self._ensure()
        # stand-alone use (not through the renderer): no weight gradients, activations are not stashed
        with torch.no_grad():
            return ops.color_fwd(self._net.blob, s.n, self.prec, normals.float().contiguous(),
                                 feature_vectors.float().contiguous(), None, False, dirs=s.dirs, **s.kw())


class SingleVarianceNetwork(nn.Module):
    """fields.py:262-268"""

    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(init_val))))

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)

    def inv_s(self):
        return torch.exp(self.variance * 10.0).clip(1e-6, 1e6).reshape(1)      # renderer.py:245


def _l2_normalize(x):
    eps = torch.finfo(torch.float32).eps
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, dim=-1, keepdim=True), min=eps))


def _linear_to_srgb(linear):
    if linear.is_cuda and linear.dtype == torch.float32:
        return ops.srgb(linear)                   # one launch (fneus_srgb_fwd / _bwd)
    eps = torch.finfo(torch.float32).eps
    srgb0 = 323.0 / 25.0 * linear
    srgb1 = (211.0 * torch.clamp(linear, min=eps) ** (5.0 / 12.0) - 11.0) / 200.0
    return torch.where(linear <= 0.0031308, srgb0, srgb1)


class _PlainBackend:
    """Device backend of one RefColor MLP: five plain nn.Linear layers whose weight / bias (and their .grad) are
    aliased onto the flat buffers of a PackedNet, exactly like _HipMLP does for the weight-normalised networks."""

    def __init__(self, kind: str, layers):
        self.kind, self.layers = kind, list(layers)
        self.ext_grad = None
        self.net = None
        self.ws = _Workspace()
        self.anchor = None
        self.packed = False

    def attached(self):
        return (self.net is not None and self.layers[0].bias.device == self.net.device
                and self.layers[0].bias.data_ptr() == self.net.raw_views(self.net.raw)[0]["bias"].data_ptr())

    def attach(self):
        dev = self.layers[0].bias.device
        if dev.type != "cuda":
            raise RuntimeError("the fneus HIP backend needs the module on a GPU (there is no CPU fallback)")
        net = ops.PackedNet(self.kind, dev, raw_grad=self.ext_grad)
        with torch.no_grad():
            for lin, view, gview in zip(self.layers, net.raw_views(net.raw), net.raw_views(net.raw_grad)):
                for name in ("bias", "weight"):
                    prm = getattr(lin, name)
                    view[name].copy_(prm.data)
                    if prm.grad is not None:
                        gview[name].copy_(prm.grad)
                    prm.data = view[name]
                    prm.grad = gview[name]
        self.net = net
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)

    def refresh(self):
        if not self.attached():
            self.attach()
            self.frozen_packed = None
        key = _frozen_key(p for lin in self.layers for p in (lin.weight, lin.bias))
        if key is not None and key == getattr(self, "frozen_packed", None) and self.packed:
            return
        self.frozen_packed = key
        for lin, gview in zip(self.layers, self.net.raw_views(self.net.raw_grad)):
            for name in ("bias", "weight"):
                prm = getattr(lin, name)
                if prm.grad is None:
                    gview[name].zero_()
                    prm.grad = gview[name]
        self.net.pack()
        self.packed = True

    def ensure(self):
        if not self.packed or not self.attached():
            self.refresh()


class RefColor(nn.Module):
    """Diffuse + specular surface colour at the two samples bracketing the first SDF sign change (fields.py:271-335).

    state_dict keys as in the reference (net_cd.*, viewdir_mlp.*, net_cs.0).  Both MLPs (net_cd; viewdir_mlp + net_cs)
    have the colour network's shape and run on the same fused HIP kernels (fneus_refcolor_fwd / _bwd); the sRGB
    transfer and the clipping are element-wise torch ops on [2B,3] tensors."""

    def __init__(self):
        super().__init__()
        self.embedview_fn, _ = get_embedder(4)
        self.net_cd = nn.Sequential(nn.Linear(286, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256),
                                    nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 3), nn.Sigmoid())
        self.viewdir_mlp = nn.ModuleList([nn.Linear(289, 256)] + [nn.Linear(256, 256) for _ in range(3)])
        self.net_cs = nn.Sequential(nn.Linear(256, 1), nn.Sigmoid())
        self.prec = ops.PREC_PARITY
        self._cd = _PlainBackend("refcd", [self.net_cd[i] for i in (0, 2, 4, 6, 8)])
        self._vd = _PlainBackend("refvd", list(self.viewdir_mlp) + [self.net_cs[0]])

    def set_precision(self, prec: int):
        assert prec in (ops.PREC_FAST, ops.PREC_PARITY)
        self.prec = prec

    def set_gradient_precision(self, gprec):
        """see SDFNetwork.set_gradient_precision"""
        assert gprec in (None, 1, 2, 3)
        self._cd.ws.gprec = gprec
        for k in [k for k in self._cd.ws.cache if k[0] in ("ref_stash", "ref_jobs")]:
            del self._cd.ws.cache[k]

    def refresh(self):
        """pack the current parameters of both MLPs (once per optimiser step, before rendering)"""
        self._cd.refresh()
        self._vd.refresh()
        # the two 32-tile launches of these heads are as long as one tile's chain and stream 1.1 MB of fragments per head:
        # name them to the launches in front of them, whose idle workgroups read them into L2 (csrc/loss_kernels.hip)
        # The ranges belong to THIS instance and travel as an argument of those launches (warm_ranges()): nothing global.
        key = (self._cd.net.blob.data_ptr(), self._vd.net.blob.data_ptr())
        if getattr(self, "_warm_key", None) != key and self._cd.net.blob.is_cuda:
            self._warm_key = key
            self._warm = (ops.warm_ranges(ops.fragment_ranges(self._cd.net, False) + ops.fragment_ranges(self._vd.net, False)),
                          ops.warm_ranges(ops.fragment_ranges(self._cd.net, True) + ops.fragment_ranges(self._vd.net, True)))

    def warm_ranges(self, backward: bool):
        """ops.WarmRanges of the packed weight fragments the next RefColor launch streams (forward / backward), or None: handed
        to the surface_gather / stage1_loss launch in front of it (fneus.h FneusWarmRanges)"""
        w = getattr(self, "_warm", None)
        return None if w is None else w[1 if backward else 0]

    def flat_grads(self):
        return [b.net.raw_grad for b in (self._cd, self._vd) if b.net is not None]

    def n_raw(self):
        return [sum(p.numel() for l in b.layers for p in l.parameters()) for b in (self._cd, self._vd)]

    def use_grad_buffers(self, buf_cd: torch.Tensor, buf_vd: torch.Tensor):
        self._cd.ext_grad, self._vd.ext_grad = buf_cd, buf_vd
        self._cd.net = self._vd.net = None
        self._cd.ws.cache.clear()      # job tables cache gradient-buffer addresses

    def heads(self, samples: RaySamples, x, n, train: bool):
        """-> diffuse [M,3], specular [M,3] with the value in column 0 (both after their sigmoid), differentiable w.r.t.
        x, n and the parameters"""
        self._cd.ensure()
        self._vd.ensure()
        return RefHeadsFn.apply(self._cd.anchor, n, x, self._cd.net, self._vd.net, samples, self.prec, self._cd.ws, train)

    @staticmethod
    def shade(diffuse, spec):
        spec = spec[:, :1].repeat(1, 3)
        if diffuse.is_cuda and diffuse.dtype == torch.float32:
            f = lambda x: ops.srgb(x, clip=True)      # clip(linear_to_srgb(.), 0, 1) in one launch
        else:
            f = lambda x: torch.clip(_linear_to_srgb(x), 0.0, 1.0)
        return {"rgb": f(spec + diffuse), "specular_rgb": f(spec), "diffuse_rgb": f(diffuse)}

    def forward_samples(self, samples: RaySamples, x, n):
        """hot-path entry: sample positions given as (rays_o, rays_d, t) like the other fused kernels"""
        return self.shade(*self.heads(samples, x, n, torch.is_grad_enabled()))

    def forward(self, pts, x, dirs, n):
        s = RaySamples(pts=pts.detach().float().contiguous(), dirs=dirs.detach().float().contiguous())
        return self.forward_samples(s, x.float(), n.float())


class NeRF(nn.Module):
    """Background NeRF++ (fields.py:178-259); only evaluated when n_outside > 0 (womask).

    Parameters are plain nn.Linear modules named as in the reference (pts_linears.{i}, views_linears.0, feature_linear,
    alpha_linear, rgb_linear); weights and gradients are aliased onto the flat buffers of a PackedNet and the forward /
    backward run on the fused HIP kernels fneus_nerf_bg_fwd / _bwd (K7).  Only the architecture of confs/womask.conf is
    supported (there is no fallback path)."""

    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, multires=0, multires_view=0, output_ch=4, skips=(4,),
                 use_viewdirs=False):
        super().__init__()
        self.D, self.W, self.skips, self.use_viewdirs = D, W, tuple(skips), use_viewdirs
        self.embed_fn, self.input_ch = (get_embedder(multires, input_dims=d_in) if multires > 0 else (None, 3))
        self.embed_fn_view, self.input_ch_view = (get_embedder(multires_view, input_dims=d_in_view)
                                                  if multires_view > 0 else (None, 3))
        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.input_ch, W)] +
            [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_view + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self.prec = ops.PREC_PARITY
        self._fused = (D == 8 and W == 256 and d_in == 4 and d_in_view == 3 and multires == 10 and multires_view == 4
                       and self.skips == (4,) and use_viewdirs)
        self._be = None
        if self._fused:       # parameter layers in the order of fneus/netdesc.py NERF_NAMES
            self._be = _PlainBackend("nerf", list(self.pts_linears) + [self.feature_linear, self.alpha_linear,
                                                                       self.views_linears[0], self.rgb_linear])

    def set_precision(self, prec: int):
        assert prec in (ops.PREC_FAST, ops.PREC_PARITY)
        self.prec = prec

    def set_gradient_precision(self, gprec):
        """1: bf16 stash planes for the weight-gradient GEMM (default), 3: hi + lo planes (fp32-accurate gradients)"""
        if self._be is not None:
            self._be.ws.gprec = gprec
            self._be.ws.cache.clear()

    def n_raw(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def use_grad_buffer(self, buf: torch.Tensor):
        self._be.ext_grad = buf
        self._be.net = None
        self._be.ws.cache.clear()      # the weight-gradient job table caches raw_grad.data_ptr()

    def refresh(self):
        """pack the current parameters (once per optimiser step, before rendering)"""
        self._be.refresh()

    def select_count(self, cap: int) -> torch.Tensor:
        """the int32 buffer that holds how many of `cap` rows a selected evaluation uses (ops.outside_select): one per size, like the
        stash -- kernels and the weight-gradient job table of the step keep its address"""
        self._be.ensure()
        return self._be.ws.get(("select_count", cap), lambda: torch.zeros(1, dtype=torch.int32, device=self._be.anchor.device))

    def forward(self, input_pts, input_views, n_active=None):
        """input_pts [N,4] inverted-sphere points, input_views [N,3] -> raw density [N,1], raw rgb [N,3]
        n_active: device int32 count (ops.outside_select): only the first n_active rows are evaluated"""
        if not self._fused:
            raise NotImplementedError("the fused background-NeRF kernels are specialised for confs/womask.conf "
                                      "(D=8, W=256, d_in=4, multires=10, multires_view=4, skips=[4], use_viewdirs)")
        self._be.ensure()
        density, rgb = NerfFn.apply(self._be.anchor, self._be.net, input_pts.detach().float().contiguous(),
                                    input_views.detach().float().contiguous(), self.prec, self._be.ws,
                                    torch.is_grad_enabled(), n_active)
        return density.reshape(-1, 1), rgb


class _IndirIllumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, dirs):
        ctx.save_for_backward(raw, dirs)
        return ops.indir_illum_fwd(raw, dirs)

    @staticmethod
    def backward(ctx, d_rad):
        raw, dirs = ctx.saved_tensors
        return ops.indir_illum_bwd(raw, dirs, d_rad.contiguous()), None


class _DirectLinearFn(torch.autograd.Function):
    """nn.Linear whose backward writes the weight / bias gradients STRAIGHT into the parameters' persistent `.grad` buffers
    (`torch.mm(..., out=weight.grad)`, `torch.sum(..., out=bias.grad)`) and hands autograd no gradient for them.  With
    persistent gradient buffers (the stage-2 / 3 trainers' gradient arena) autograd's own route is mm -> temporary ->
    `grad += temporary` per tensor: two launches more per layer on tensors of a few hundred KB.  Valid when each parameter is
    used once per backward and its buffer is overwritten, not accumulated into (the trainers clear the arena every step)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x)
        ctx.weight, ctx.bias = weight, bias
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, b = ctx.weight, ctx.bias
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        torch.mm(dy2.t(), x2, out=w.grad)
        torch.sum(dy2, 0, out=b.grad)
        dx = (dy2 @ w.detach()).reshape(x.shape) if ctx.needs_input_grad[0] else None
        return dx, None, None


def _mlp_spec(seq):
    """[(Linear, activation code)] of an nn.Sequential of nn.Linear layers each followed by at most one of ReLU /
    LeakyReLU(0.2) / Sigmoid (what fneus_mlp_* evaluates), else None"""
    layers = []
    for m in seq:
        if isinstance(m, nn.Linear):
            layers.append([m, ops.ACT_NONE])
        elif not layers or layers[-1][1] != ops.ACT_NONE:
            return None
        elif isinstance(m, nn.ReLU):
            layers[-1][1] = ops.ACT_RELU
        elif isinstance(m, nn.LeakyReLU) and abs(m.negative_slope - 0.2) < 1e-12:
            layers[-1][1] = ops.ACT_LEAKY02
        elif isinstance(m, nn.Sigmoid):
            layers[-1][1] = ops.ACT_SIGMOID
        else:
            return None
    return layers or None


class _MlpGroupFn(torch.autograd.Function):
    """Independent nn.Sequential MLPs of Linear + activation layers IN LOCKSTEP on the fneus_mlp_* kernels
    (csrc/mlp_rows_kernels.hip): layer p of every network is ONE launch forward, one for the input gradients, and every weight and
    bias gradient of all of them one launch (16 layers per launch) -- through torch a GEMM per Linear and direction, an
    element-wise launch per activation and direction and a reduction per bias gradient, per network.
    args: nets = ((activation codes, direct), ...), then per network x [..., n_in] and weight, bias per layer (bias may be None).
    direct: that network's parameter gradients are written STRAIGHT into the parameters' persistent `.grad` buffers (overwritten) and
    autograd gets none for them -- the stage-2 / 3 trainers' gradient arena, see _DirectLinearFn.  -> one output per network."""

    @staticmethod
    def _split(nets, tensors):
        pos, out = 0, []
        for acts, _ in nets:
            out.append((pos, tensors[pos], tensors[pos + 1:pos + 1 + 2 * len(acts)]))
            pos += 1 + 2 * len(acts)
        return out

    @staticmethod
    def forward(ctx, nets, *tensors):
        parts = _MlpGroupFn._split(nets, tensors)
        x2s = [x.reshape(-1, x.shape[-1]).contiguous() for _, x, _ in parts]
        ys = [[] for _ in nets]
        for p in range(max(len(acts) for acts, _ in nets)):
            jobs = []
            for n, ((acts, _), (_, x, wb)) in enumerate(zip(nets, parts)):
                if p >= len(acts):
                    continue
                w, b = wb[2 * p], wb[2 * p + 1]
                rows = x2s[n].shape[0]
                y = torch.empty(rows, w.shape[0], dtype=torch.float32, device=x.device)
                if rows:
                    jobs.append(dict(x=ys[n][-1] if p else x2s[n], weight=w.detach(), bias=None if b is None else b.detach(), y=y,
                                     rows=rows, n_in=w.shape[1], n_out=w.shape[0], act=acts[p]))
                ys[n].append(y)
            for i in range(0, len(jobs), ops.MLP_MAX_JOBS):
                ops.mlp_forward(jobs[i:i + ops.MLP_MAX_JOBS])
        ctx.save_for_backward(*x2s, *[y for net in ys for y in net])
        ctx.nets, ctx.parts, ctx.shapes = nets, [(pos, wb) for pos, _, wb in parts], [x.shape for _, x, _ in parts]
        outs = tuple(net[-1].reshape(*x.shape[:-1], net[-1].shape[-1]) for net, (_, x, _) in zip(ys, parts))
        # a frozen network on a constant input (stage 3's IndirectLight riding along with the trained ones): nothing to differentiate
        frozen = [o for o, (_, x, wb) in zip(outs, parts) if not any(t is not None and t.requires_grad for t in (x, *wb))]
        if frozen:
            ctx.mark_non_differentiable(*frozen)
        ctx.set_materialize_grads(False)       # (no zero fill for the cotangent of an output nobody differentiates)
        return outs

    @staticmethod
    def backward(ctx, *douts):
        nets, N = ctx.nets, len(ctx.nets)
        saved = list(ctx.saved_tensors)
        x2s, flat, ys = saved[:N], saved[N:], []
        for acts, _ in nets:
            ys.append(flat[:len(acts)])
            flat = flat[len(acts):]
        grads = [None] * (sum(1 + 2 * len(acts) for acts, _ in nets))
        need = lambda i: ctx.needs_input_grad[1 + i]
        delta = [[None] * len(acts) for acts, _ in nets]
        for n, (acts, _) in enumerate(nets):
            pos, wb = ctx.parts[n]
            if douts[n] is not None:
                delta[n][-1] = douts[n].reshape(x2s[n].shape[0], ys[n][-1].shape[1]).contiguous()
            elif any(need(pos + i) for i in range(1 + len(wb))):       # an unused output of a network that is differentiated
                delta[n][-1] = torch.zeros_like(ys[n][-1])
        top = lambda n, l: nets[n][0][l] if l == len(nets[n][0]) - 1 else ops.ACT_NONE   # below the top layer dy is the pre-activation's gradient
        # input gradients, top layers first: step t takes layer L - 1 - t of every network, down to the lowest layer that still
        # has something to differentiate below it (`low`: 0 = the input itself; L = nothing at all, a frozen network)
        low = []
        for n, (acts, _) in enumerate(nets):
            pos, wb = ctx.parts[n]
            wanted = [l for l in range(len(acts)) if need(pos + 1 + 2 * l) or (wb[2 * l + 1] is not None and need(pos + 2 + 2 * l))]
            low.append(-1 if need(pos) else (wanted[0] if wanted else len(acts)))
        for t in range(max(len(acts) for acts, _ in nets)):
            jobs, outs = [], []
            for n, (acts, _) in enumerate(nets):
                l = len(acts) - 1 - t
                pos, wb = ctx.parts[n]
                rows = x2s[n].shape[0]
                if l < 0 or l <= low[n]:
                    continue
                w = wb[2 * l].detach()
                dx = torch.empty(rows, w.shape[1], dtype=torch.float32, device=w.device)
                if rows:
                    jobs.append(dict(dy=delta[n][l], y=ys[n][l] if top(n, l) else None, weight=w, x=ys[n][l - 1] if l else None, dx=dx,
                                     rows=rows, n_in=w.shape[1], n_out=w.shape[0], act=top(n, l), act_in=acts[l - 1] if l else 0))
                if l:
                    delta[n][l - 1] = dx
                else:
                    grads[pos] = dx.reshape(ctx.shapes[n])
            for i in range(0, len(jobs), ops.MLP_MAX_JOBS):
                ops.mlp_backward_input(jobs[i:i + ops.MLP_MAX_JOBS])
        jobs = []
        for n, (acts, direct) in enumerate(nets):
            pos, wb = ctx.parts[n]
            rows = x2s[n].shape[0]
            for l in range(len(acts)):
                w, b = wb[2 * l], wb[2 * l + 1]
                need_w, need_b = need(pos + 1 + 2 * l), b is not None and need(pos + 2 + 2 * l)
                if not (need_w or need_b):
                    continue
                if direct:
                    dw, db = w.grad, (b.grad if need_b else None)
                else:
                    dw = torch.empty_like(w, memory_format=torch.contiguous_format)
                    db = torch.empty_like(b) if need_b else None
                    grads[pos + 1 + 2 * l], grads[pos + 2 + 2 * l] = (dw if need_w else None), db
                if rows:
                    jobs.append(dict(dy=delta[n][l], y=ys[n][l] if top(n, l) else None, x=ys[n][l - 1] if l else x2s[n], d_weight=dw,
                                     d_bias=db, rows=rows, n_in=w.shape[1], n_out=w.shape[0], act=top(n, l)))
                else:                 # no rows: the sums are empty
                    dw.zero_()
                    if db is not None:
                        db.zero_()
        for i in range(0, len(jobs), ops.MLP_MAX_JOBS):
            ops.mlp_backward_params(jobs[i:i + ops.MLP_MAX_JOBS])
        return (None, *grads)


_TOP_ACTS = {ops.ACT_NONE: lambda x: x, ops.ACT_RELU: torch.relu, ops.ACT_LEAKY02: lambda x: torch.nn.functional.leaky_relu(x, 0.2),
             ops.ACT_SIGMOID: torch.sigmoid}


def seq_group(items):
    """[(nn.Sequential, x, owner[, top_act]), ...] -> [outputs]: top_act (an ops.ACT_* code) is an activation the CALLER applies to
    the output of a network that ends in a Linear layer (torch.sigmoid(decoder(x))): it becomes the last layer's activation.
    [(nn.Sequential, x, owner), ...] -> [outputs]: independent MLPs of Linear + activation layers evaluated in lockstep.  On the
    GPU: the fneus_mlp_* kernels (_MlpGroupFn; FNEUS_MLP_ROWS=0 keeps torch's modules), a network's parameter gradients written
    straight into persistent `.grad` buffers when its owner asks for it (`owner.direct_grads`, set by the trainers that keep a
    gradient arena, for the duration of their own steps) and every buffer exists.  Networks the kernels do not take (another
    activation, another dtype, the CPU) run as the plain modules, Linear layers through _DirectLinearFn under the same condition."""
    def has_buffers(m):
        return m.weight.requires_grad and m.weight.grad is not None and m.bias is not None and m.bias.grad is not None \
            and m.weight.grad.is_contiguous()

    results = [None] * len(items)
    nets, tensors, where = [], [], []
    for idx, item in enumerate(items):
        seq, x, owner = item[:3]
        top_act = item[3] if len(item) > 3 else None
        want_direct = getattr(owner, "direct_grads", False) and torch.is_grad_enabled()
        spec = _mlp_spec(seq) if (ops.MLP_ROWS and x.is_cuda and x.dtype == torch.float32) else None
        if spec is not None and x.numel() // max(1, x.shape[-1]) * max(max(m.in_features, m.out_features) for m, _ in spec) * 4 >= 2 ** 31:
            spec = None            # (the kernels address an operand through 32-bit offsets: such a batch goes to torch's modules)
        if spec is not None:
            # what nn.Linear would reject: an input whose width is not the first layer's, or consecutive layers that do not fit -- the
            # kernels would read rows of the wrong stride (the plain modules raise the error torch's users know)
            widths_ok = x.shape[-1] == spec[0][0].in_features and all(a.out_features == b.in_features for (a, _), (b, _) in zip(spec, spec[1:]))
            if not widths_ok:
                spec = None
        if spec is not None and top_act is not None:
            if spec[-1][1] != ops.ACT_NONE:
                raise ValueError("seq_group: top_act given for a network that ends in an activation")
            spec[-1][1] = top_act
        if spec is not None and all(m.weight.dtype == torch.float32 and m.weight.is_contiguous() for m, _ in spec):
            direct = bool(want_direct) and all(has_buffers(m) for m, _ in spec)
            nets.append((tuple(a for _, a in spec), direct))
            tensors += [x] + [p for m, _ in spec for p in (m.weight, m.bias)]
            where.append(idx)
        else:
            if not want_direct:
                x = seq(x)
            else:
                for m in seq:
                    x = _DirectLinearFn.apply(x, m.weight, m.bias) if isinstance(m, nn.Linear) and has_buffers(m) else m(x)
            results[idx] = x if top_act is None else _TOP_ACTS[top_act](x)
    if nets:
        for idx, y in zip(where, _MlpGroupFn.apply(tuple(nets), *tensors)):
            results[idx] = y
    return results


def _seq_direct(seq, x, owner):
    """one nn.Sequential through seq_group"""
    return seq_group([(seq, x, owner)])[0]


class Lvis(nn.Module):
    """Stage-2 distilled light visibility (fields.py:338-369): sigmoid(MLP(embed(pts, 10) | embed(view, 4))), 90 -> 256 x 4 -> 1.
    Same state_dict keys as the reference (`lvis.{0,2,4,6,8}.weight / bias`); the first layer is an explicit Linear(90, 256)
    where the reference uses LazyLinear.  A plain library MLP: the GEMMs go to rocBLAS through torch (<= 2048 rows per step,
    < 2 % of the stage-2 step, which is the 1 M-point SDF march on the K1 kernel)."""

    def __init__(self):
        super().__init__()
        self.embedview_fn_view, ch_view = get_embedder(4)
        self.embedview_fn_pts, ch_pts = get_embedder(10)
        self.lvis = nn.Sequential(nn.Linear(ch_pts + ch_view, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                  nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 1),
                                  nn.Sigmoid())

    def mlp_input(self, pts, view):
        return torch.cat([self.embedview_fn_pts(pts), self.embedview_fn_view(view)], dim=-1)

    def forward(self, pts, view):
        return _seq_direct(self.lvis, self.mlp_input(pts, view), self)

    def visibility(self, points, normals, dirs, weights, point_mask=None):
        """get_diffuse_visibility's network part (inverRender.py:163-190), no gradient: for every surface point the network
        at the S directions of each of the M light lobes, zero where the direction faces away from the normal, averaged per
        lobe with the given weights.  points, normals [n,3]; dirs [M,S,3]; weights [M,S] -> [M,n].  point_mask [n] bool: the
        points marked False are not evaluated (visibility 0): the fixed-shape stage-3 step marks the rays that hit"""
        if dirs.shape[1] != 32:          # the fused kernel maps one (point, lobe) pair onto one 32-sample MFMA tile
            vis = self._visibility_library(points, normals, dirs, weights)
            return vis if point_mask is None else vis * point_mask[None, :].to(vis.dtype)
        net = self._packed()
        blob = net.blob
        if self.prec == ops.PREC_H16:                # the blob with one fp16 value per weight, converted when the packed blob changed
            if getattr(self, "_h16_key", None) != self._pack_key:
                self._h16_blob, self._h16_key = ops.lvis_h16_pack(net.blob), self._pack_key
            blob = self._h16_blob
        return ops.lvis_visibility(blob, points, normals, dirs.contiguous(), weights.contiguous(), self.prec, point_mask)

    # The visibility launch of stage 3 in the 1e-4 mode: ops.PREC_PARITY (three bf16 products; a lobe's visibility within 3e-7 of
    # float64) or ops.PREC_H16 (ONE fp16 product: within 3e-5 -- a lobe averages up to 32 sigmoid outputs -- at half the time:
    # 1.27 -> 0.63 ms per stage-3 step).  FNEUS_LVIS_PREC selects the default.
    prec = int(os.environ.get("FNEUS_LVIS_PREC", str(ops.PREC_PARITY)))

    def set_precision(self, prec: int):
        assert prec in (ops.PREC_FAST, ops.PREC_H16, ops.PREC_PARITY)
        self.prec = prec

    def _packed(self):
        """the weights as MFMA fragments (fneus_pack, layout 3); re-packed when a parameter has changed (stage 3 keeps this
        network frozen, so that is once)"""
        layers = [self.lvis[i] for i in (0, 2, 4, 6, 8)]
        dev = layers[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("the fneus HIP backend needs the module on a GPU (there is no CPU fallback)")
        key = tuple((p.data_ptr(), p._version) for lin in layers for p in (lin.weight, lin.bias))
        if getattr(self, "_pack_key", None) != key:
            net = getattr(self, "_net", None)
            if net is None or net.device != dev:
                net = ops.PackedNet("lvis", dev)
            with torch.no_grad():
                for lin, view in zip(layers, net.raw_views(net.raw)):
                    view["weight"].copy_(lin.weight)
                    view["bias"].copy_(lin.bias)
            net.pack()
            self._net, self._pack_key = net, key
        return self._net

    @torch.no_grad()
    def _visibility_library(self, points, normals, dirs, weights, chunk: int = 32):
        """the same through the library GEMMs (rocBLAS), `chunk` points x M*S directions per pass"""
        M, S = weights.shape
        flat = dirs.reshape(M * S, 3)
        d_enc = self.embedview_fn_view(flat)                                  # [MS,27]: the same for every point
        wsum = weights.sum(dim=1) + 1e-6
        out = torch.empty(M, points.shape[0], device=points.device)
        for i in range(0, points.shape[0], chunk):
            p, nrm = points[i:i + chunk], normals[i:i + chunk]
            c = p.shape[0]
            front = (nrm @ flat.t()) > 1e-6                                   # [c, MS]
            x = torch.cat([self.embedview_fn_pts(p)[:, None, :].expand(c, M * S, -1), d_enc[None].expand(c, -1, -1)], dim=-1)
            vis = self.lvis(x.reshape(c * M * S, -1)).reshape(c, M, S) * front.reshape(c, M, S)
            out[:, i:i + c] = ((vis * weights[None]).sum(dim=2) / wsum[None]).t()
        return out


class DeferredIndirectLight:
    """IndirectLight(pts) that has not run yet (IndirectLight.deferred)"""

    def __init__(self, net, pts):
        self.net, self.pts = net, pts

    def resolve(self):
        with torch.no_grad():
            return self.net(self.pts)


class IndirectLight(nn.Module):
    """Stage-2 indirect light as 24 spherical Gaussians per point (fields.py:372-413): 63 -> 512 x 4 -> 144 -> [n, 24, 7] =
    (lobe axis from two sigmoid angles, sharpness 30 sigmoid + 0.1, relu amplitude x 3).  state_dict keys `indi.{0,..,8}`."""

    def __init__(self, num_lgt_sgs=24):
        super().__init__()
        self.num_lgt_sgs = num_lgt_sgs
        self.embedview_fn_view, _ = get_embedder(4)
        self.embedview_fn_pts, ch_pts = get_embedder(10)
        self.indi = nn.Sequential(nn.Linear(ch_pts, 512), nn.ReLU(), nn.Linear(512, 512), nn.ReLU(), nn.Linear(512, 512),
                                  nn.ReLU(), nn.Linear(512, 512), nn.ReLU(), nn.Linear(512, num_lgt_sgs * 6))

    def radiance(self, pts, sample_dirs):
        """query_indir_illum(self(pts), sample_dirs) (calLvis.py:323-336) [n, S, 3] without materialising the lobes: the output
        transform below and the sum of spherical Gaussians in one launch forward, one backward (fneus_indir_illum_fwd / _bwd)"""
        return self.radiance_from_raw(_seq_direct(self.indi, self.embedview_fn_pts(pts), self), sample_dirs)

    def radiance_from_raw(self, raw, sample_dirs):
        """raw: the MLP's output [n, 6 L] (`self.indi` on `self.embedview_fn_pts(pts)`)"""
        raw = raw.reshape(-1, self.num_lgt_sgs, 6)
        return _IndirIllumFn.apply(raw.contiguous(), sample_dirs.detach().float().contiguous())

    def deferred(self, pts):
        """the lobes of `pts`, not evaluated yet: a caller that runs other MLPs on the same points (EnvmapMaterialNetwork.forward)
        takes this network's layers into the same launches (seq_group) and finishes with sgs_from_raw"""
        return DeferredIndirectLight(self, pts)

    def sgs_from_raw(self, raw):
        """the output transform (fields.py:395-413) of the MLP's output [n, 6 L], no gradient -> [n, L, 7]"""
        return ops.indir_sgs(raw.detach().reshape(-1, self.num_lgt_sgs, 6).contiguous())

    def forward(self, pts):
        out = _seq_direct(self.indi, self.embedview_fn_pts(pts), self).reshape(-1, self.num_lgt_sgs, 6)
        if out.is_cuda and out.dtype == torch.float32 and not (torch.is_grad_enabled() and out.requires_grad):
            return ops.indir_sgs(out)             # frozen network (stage 3): the lines below in one launch
        ang = torch.sigmoid(out[..., :2]) * (2 * np.pi)
        theta, phi = ang[..., :1], ang[..., 1:2]
        lobes = torch.cat([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], dim=-1)
        lam = torch.sigmoid(out[..., 2:3]) * 30 + 0.1
        mu = torch.relu(out[..., 3:])
        return torch.cat([lobes, lam, mu], dim=-1)
