"""torch.autograd glue between the HIP ops (torch is plumbing: it owns memory, streams and the graph).

Each Function wraps one forward kernel and its hand-written backward kernels; the maths lives in csrc/.
Points are never differentiated (all z values are produced under no_grad in the reference, renderer.py:188, 426).
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import ops


class RaySamples:
    """Sample positions p = rays_o[n // m] + rays_d[n // m] * t[n]  (or explicit pts)."""

    def __init__(self, rays_o=None, rays_d=None, t=None, m: int = 1, pts=None, dirs=None):
        self.rays_o, self.rays_d, self.t, self.m, self.pts, self.dirs = rays_o, rays_d, t, m, pts, dirs
        self.n = int(pts.shape[0]) if pts is not None else int(t.numel())

    def kw(self):
        return dict(pts=self.pts, rays_o=self.rays_o, rays_d=self.rays_d, t=self.t, m=self.m)


class _Workspace:
    """Per-network cache of stash / work buffers keyed by (n, prec): allocated once, reused every step."""

    def __init__(self):
        self.cache = {}
        self.gprec = None          # gradient precision of the backward stash (ops.DEFAULT_GPREC when None)

    def get(self, key, factory):
        if key not in self.cache:
            self.cache[key] = factory()
        return self.cache[key]

    # The stash / work buffers are shared per (n, prec): ONE differentiable call per network and size may be in flight.
    # A second forward of the same size before the first one's backward would overwrite its stash and the first backward
    # would silently use the wrong activations, so every forward stamps its stash and the backward checks the stamp.
    def stamp(self, stash):
        self.generation = getattr(self, "generation", 0) + 1
        stash.generation = self.generation
        return self.generation

    @staticmethod
    def check(stash, generation, what):
        if getattr(stash, "generation", None) != generation:
            raise RuntimeError(f"{what}: the activation stash of this call was overwritten by a later forward of the same size "
                               "before its backward ran (the fused networks keep one stash per batch size: call backward "
                               "before the next differentiable forward, or use different batch sizes)")


class SdfValueGradFn(torch.autograd.Function):
    """K2 forward; backward = K3 + weight-gradient GEMM + weight-norm backward.  The parameters are not autograd
    inputs: their gradients are ACCUMULATED into the network's flat gradient buffer (which the Parameters' .grad
    alias); `anchor` is a dummy leaf that makes autograd call backward."""

    @staticmethod
    def forward(ctx, anchor, net, samples: RaySamples, prec: int, ws: _Workspace, train: bool, feat_rows: bool = True):
        """feat_rows False (NeuSRenderer.render_core, whose consumers of the feature vector read the stash's planes: round 6): where the
        launch allows it (ops.feat_planes_ok) the returned feature tensor is a placeholder no kernel wrote"""
        n = samples.n
        stash = ws.get(("sdf_stash", n, prec, train), lambda: ops.SdfStash(n, anchor.device, prec, train, gprec=ws.gprec))
        rows = feat_rows or not ops.feat_planes_ok(n, prec, train)
        sdf, feat, normal = ops.sdf_fwd_grad(net.blob, n, prec, stash, train, feat_rows=rows, **samples.kw())
        ctx.net, ctx.samples, ctx.prec, ctx.ws, ctx.stash, ctx.n = net, samples, prec, ws, stash, n
        ctx.generation = ws.stamp(stash)
        if train:       # the forward whose backward is still to come (its stamp, not a count: a forward whose backward never runs
            _PENDING_OPEN[anchor.device] = (id(ws), ctx.generation)       # is replaced by the next one)
        return sdf, feat, normal

    @staticmethod
    def backward(ctx, d_sdf, d_feat, d_normal):
        n, prec, net, ws = ctx.n, ctx.prec, ctx.net, ctx.ws
        ws.check(ctx.stash, ctx.generation, "SDFNetwork backward")
        # every consumer of sdf / feature / normal has run its backward by now: the gradients of the colour network, the
        # RefColor heads and the variance are final.  The data-parallel trainer starts their exchange here, beside K3.
        dev = d_feat.device if d_feat is not None else (d_sdf.device if d_sdf is not None else d_normal.device)
        ops.flush_fold_rider(dev)          # (a variance-gradient sum no fold launch has taken along: gradient precision 1 / 3)
        hook = getattr(ws, "pre_backward", None)
        if hook is not None:
            hook()
        d_sdf = torch.zeros(n, device=dev) if d_sdf is None else d_sdf.contiguous()
        d_feat = torch.zeros(n, 256, device=dev) if d_feat is None else d_feat.contiguous()
        d_normal = torch.zeros(n, 3, device=dev) if d_normal is None else d_normal.contiguous()
        bufs = ws.get(("sdf_bwd", n, prec), lambda: ops.SdfBwdBufs(n, dev, prec, gprec=ctx.stash.gprec))
        # Round 6: ColorFn.backward has written the feature cotangent straight into this kernel's seed plane (bf16 fragments, slot 8 of
        # zbar) and filed the fact under this forward's stamp; the tensor autograd delivered is then its placeholder.  Anything else
        # (another consumer of the features summed in by autograd, a record of another forward) falls back to rows.
        plane = ws.cache.pop("dfeat_in_plane", None)
        in_plane = (plane is not None and plane[0] == (id(ws), ctx.generation) and getattr(d_feat, "plane_of", None) is not None
                    and d_feat.plane_of.data_ptr() == bufs.zbar[0, 8].data_ptr())
        if plane is not None and not in_plane:
            if getattr(d_feat, "plane_of", None) is not None or plane[0] == (id(ws), ctx.generation):
                raise RuntimeError("SDFNetwork backward: the feature cotangent was left in a fragment plane this backward cannot use "
                                   "(FNEUS_DFEAT_PLANE=0 keeps fp32 rows)")
        # gradients of the two gathered surface samples per ray arrive on the side (SurfaceGatherFn): 2B rows to add
        # instead of a dense, mostly zero [n,256] tensor for autograd to allocate and sum
        pending = ws.cache.pop("surface_grads", None)
        if pending is not None:
            sel, dfs, dns = pending
            if dfs is not None:
                if in_plane:
                    ops.surface_scatter_plane(sel, dfs[None].contiguous(), None, bufs.zbar[0, 8], n, d_normal)
                else:
                    d_feat.index_add_(0, sel, dfs)
            if dns is not None:
                d_normal.index_add_(0, sel, dns)
        # RefHeadsFn.backward: per-head gradients of the gathered rows, filed under the stamp of the SDF forward the rows came from (two
        # renders before one backward pass each find their own record, in whatever order autograd runs the nodes)
        heads = ws.cache.get("surface_head_grads", {}).pop((id(ws), ctx.generation), None)
        if heads is not None:
            if in_plane:
                ops.surface_scatter_plane(heads[0], heads[1], heads[2], bufs.zbar[0, 8], n, d_normal)
            else:
                ops.surface_scatter(heads[0], heads[1], heads[2], d_feat, d_normal)       # head sum + scatter-add in one launch
        ops.sdf_bwd(net.blob, n, prec, ctx.stash, bufs, d_sdf, None if in_plane else d_feat, d_normal, **ctx.samples.kw())
        # zeroed once: fneus_wn_backward clears what it reads, so the buffer is zero again after every step
        grad = ws.get(("sdf_grad", n), lambda: torch.zeros(net.n_params, dtype=torch.float32, device=dev))
        # the colour network's products over the same samples wait here (ColorFn.backward): one launch for both networks
        # products that wait for this launch: only records made for THIS forward (a backward pass that died half way leaves its
        # records behind: they must not ride in a later step's launch)
        mine = (id(ws), ctx.generation)
        if _PENDING_OPEN.get(dev) == mine:
            _PENDING_OPEN[dev] = None
        col, ref = ws.cache.pop("pending_color_dw", None), ws.cache.pop("pending_ref_dw", None)
        # a record made for ANOTHER forward of this workspace (two renders before one backward pass) is not dropped: its
        # products run as a launch of their own, unless its own stash has been overwritten since (then it is stale)
        if col is not None and col["for"] != mine:
            if _record_alive(col["stash"], col["gen"]):
                _run_color_dw(col)
            col = None
        if ref is not None and ref["for"] != mine:
            if _record_alive(ref["st"][0], ref["gen"]):
                _run_ref_dw(ref)
            ref = None
        bgs = []
        for r in _PENDING_NERF.pop(dev, {}).values():
            if r["for"] == mine:
                bgs.append(r)
            elif _record_alive(r["stash"], r["gen"]):
                _run_nerf_dw(r)
        # (plane precision: gradient precision 2 is hi planes for everything that rides in this launch)
        planes = lambda st: 1 if st.gprec == 2 else st.gprec
        if col is not None and (col["n"] != n or planes(col["stash"]) != planes(ctx.stash)):
            _run_color_dw(col)
            col = None
        if ref is not None and (col is None or planes(ref["st"][0]) != planes(ctx.stash)):
            _run_ref_dw(ref)
            ref = None
        for r in [r for r in bgs if col is None or planes(r["stash"]) != planes(ctx.stash)]:
            _run_nerf_dw(r)
            bgs.remove(r)
        if col is not None:
            def also(g):
                ops.color_dw_jobs(col["net"], col["feat_planes"], col["stash"], col["grad"], n, into=g)
                if ref is not None:      # the heads' planes hold the 2 B gathered rows: products with their own tile count
                    for net_h, st_h, g_h in zip(ref["nets"], ref["st"], ref["grads"]):
                        ops.color_dw_jobs(net_h, st_h.feat, st_h, g_h, ref["n"], into=g, own_tiles=True)
                for r in bgs:            # the background network's: their own tile count and device-side sample count
                    ops.nerf_dw_jobs(r["net"], r["stash"], r["n"], into=g, n_dev=r["n_dev"])

            jobs = ws.get(("sdf_col_jobs", n, prec) + col["key"] + (ref["key"] if ref is not None else ()) + sum((r["key"] for r in bgs), ()),
                          lambda: ops.sdf_dw_jobs(net, ctx.stash, bufs, grad, n, also=also))
            jobs.run()
            col["net"].wn_backward(col["grad"])
            if ref is not None:
                for net_h, g_h in zip(ref["nets"], ref["grads"]):
                    net_h.wn_backward(g_h)
        else:
            jobs = ws.get(("sdf_jobs", n, prec), lambda: ops.sdf_dw_jobs(net, ctx.stash, bufs, grad, n))
            jobs.run()
        net.wn_backward(grad)
        return None, None, None, None, None, None, None


def _record_alive(stash, gen):
    """a waiting weight-gradient record may still run iff the stash it reads is the one its backward wrote into"""
    return stash is not None and getattr(stash, "generation", None) == gen


def _run_ref_dw(ref):
    """the RefColor heads' weight-gradient products as a launch of their own"""
    ref["ws"].get(("ref_jobs", ref["n"], ref["prec"]), ref["build"]).run()
    for net, g in zip(ref["nets"], ref["grads"]):
        net.wn_backward(g)


def _run_color_dw(col):
    """the colour network's weight-gradient products as a launch of their own"""
    ws, n, prec, fp = col["ws"], col["n"], col["prec"], col["feat_planes"]
    # the job table holds raw device pointers into the feature planes: keyed by their address, so that a stash re-created
    # by SDFNetwork.set_gradient_precision / use_grad_buffer cannot leave a table behind that reads freed memory
    jobs = ws.get(("col_jobs", n, prec) + col["key"], lambda: ops.color_dw_jobs(col["net"], fp, col["stash"], col["grad"], n))
    jobs.run()
    col["net"].wn_backward(col["grad"])


class ColorFn(torch.autograd.Function):
    """K4 forward / backward (+ weight-gradient GEMM) of a colour-shaped MLP: the colour network (head 0, features
    stashed by the SDF kernel) or one of the two RefColor MLPs (head 1 / 2, features stashed by the kernel itself).
    Differentiable inputs: normal, feature; parameter gradients are accumulated into the network's flat buffer."""

    @staticmethod
    def forward(ctx, anchor, normal, feat, net, samples: RaySamples, prec: int, ws: _Workspace, sdf_ws: _Workspace,
                train: bool, head: int = 0):
        n = samples.n
        normal, feat = normal.contiguous(), feat.contiguous()
        stash = ws.get(("col_stash", n, prec), lambda: ops.ColStash(n, anchor.device, prec, with_feat=head != 0,
                                                                    gprec=ws.gprec)) if train else None
        rgb = ops.color_fwd(net.blob, n, prec, normal, feat, stash, train, dirs=samples.dirs, head=head, **samples.kw())
        ctx.net, ctx.prec, ctx.ws, ctx.sdf_ws, ctx.stash, ctx.n, ctx.head, ctx.samples = net, prec, ws, sdf_ws, stash, n, head, samples
        ctx.generation = ws.stamp(stash) if stash is not None else None
        # the SDF forward whose feature planes this call consumes: the latest one on that workspace NOW (by the time of the
        # backward a later render may have stamped it again)
        ctx.sdf_generation = getattr(sdf_ws, "generation", None) if sdf_ws is not None else None
        ctx.from_planes = head == 0 and getattr(feat, "planes_of", None) is not None
        ctx.normal_key = (normal.data_ptr(), tuple(normal.shape)) if head == 0 else None
        if head != 0:
            ctx.save_for_backward(rgb, normal)
        else:
            ctx.save_for_backward(rgb)
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        n, prec, net, ws, head = ctx.n, ctx.prec, ctx.net, ctx.ws, ctx.head
        ws.check(ctx.stash, ctx.generation, "colour network backward")
        if head != 0:
            rgb, normal = ctx.saved_tensors
            sm = ctx.samples
            d_feat, d_normal = ops.color_bwd(net.blob, n, prec, d_rgb.contiguous(), rgb, ctx.stash, head=head, normal=normal,
                                             dirs=sm.dirs, rays_d=sm.rays_d, m=sm.m)
            feat_planes = ctx.stash.feat
        else:
            (rgb,) = ctx.saved_tensors
            d_rgb = d_rgb.contiguous()
            # the feature cotangent as the bf16 fragments the SDF network's backward seeds its descending chain with (round 6): where the
            # features came to this call as the SDF stash's planes (their only dense consumer is this network, NeuSRenderer.render_core)
            # and both backward kernels run on bf16 cotangents, no fp32 rows [n, 256] are written or read
            plane = None
            sdf_stash = ctx.sdf_ws.cache.get(("sdf_stash", n, prec, True)) if ctx.sdf_ws is not None else None
            if (ctx.from_planes and ctx.needs_input_grad[2] and sdf_stash is not None and getattr(sdf_stash, "generation", None) == ctx.sdf_generation
                    and ctx.stash.gprec == sdf_stash.gprec and ops.dfeat_plane_ok(n, prec, sdf_stash.gprec)):
                sdf_ws = ctx.sdf_ws
                bufs = sdf_ws.get(("sdf_bwd", n, prec), lambda: ops.SdfBwdBufs(n, rgb.device, prec, gprec=sdf_stash.gprec))
                plane = bufs.zbar[0, 8]
                sdf_ws.cache["dfeat_in_plane"] = ((id(sdf_ws), ctx.sdf_generation),)
            # the compositing backward has left ITS gradient of these normals for this launch to add into (round 6): autograd's sum of
            # the two without the launch that forms it
            acc = _DN_ACC.pop(ctx.normal_key, None) if ctx.normal_key is not None else None
            if acc is not None and not (ctx.needs_input_grad[1] and ops.dnormal_accum_ok(n, prec)):
                acc = None
            d_feat, d_normal = ops.color_bwd(net.blob, n, prec, d_rgb, rgb, ctx.stash, dfeat_plane=plane, dn_accum=acc)
            feat_planes = ctx.sdf_ws.cache[("sdf_stash", n, prec, True)].feat
        grad = ws.get(("col_grad", n), lambda: torch.zeros(net.n_params, dtype=torch.float32, device=rgb.device))
        if ctx.stash.gprec == 2:        # the output layer's product with exact operands (u_3 hi + lo from the forward, zout in fp32 from
            ops.color_out_dw(net, ctx.stash, d_rgb, rgb, grad, n, cache=ws)       # d_rgb and rgb); the others follow in the merged launch as usual
        col = dict(ws=ws, net=net, n=n, prec=prec, stash=ctx.stash, grad=grad, feat_planes=feat_planes, gen=ctx.generation,
                   **{"for": (id(ctx.sdf_ws), ctx.sdf_generation)},
                   key=(feat_planes.data_ptr(), tuple(feat_planes.shape), ctx.stash.zbar.data_ptr(), grad.data_ptr()))
        # Head 0 reads the SDF network's feature planes, so the SDF network's backward follows in this backward pass whenever the
        # features carry a gradient: its weight-gradient launch takes these products along (same samples, one launch instead of
        # two: the small one ran at 2.7 TB/s, the big one runs at 4.4).  Not when somebody needs the colour gradients before
        # that launch (the data-parallel step's early exchange), and whatever is still waiting when the backward pass ends is
        # run then.
        early = getattr(ctx.sdf_ws, "color_grads_early", None)
        if head == 0 and ctx.needs_input_grad[2] and ops.gemm_merge_enabled() and not (early is not None and early()):
            sdf_ws = ctx.sdf_ws
            sdf_ws.cache["pending_color_dw"] = col

            def flush():
                if sdf_ws.cache.get("pending_color_dw") is col:
                    _run_color_dw(sdf_ws.cache.pop("pending_color_dw"))

            torch.autograd.Variable._execution_engine.queue_callback(flush)
        else:
            with ops.on_side_stream(4 if head == 0 else 2):
                _run_color_dw(col)
        return None, d_normal, d_feat, None, None, None, None, None, None, None


class NerfFn(torch.autograd.Function):
    """K7: the background NeRF++ (fields.py:233-259).  Its inputs are constants (sample positions and view directions);
    the backward writes the dL/dz planes and accumulates the parameter gradients into the network's flat buffer."""

    @staticmethod
    def forward(ctx, anchor, net, pts4, dirs, prec: int, ws: _Workspace, train: bool, n_dev=None):
        """n_dev: device int32 count (ops.outside_select): rows beyond it are not evaluated"""
        n = int(pts4.shape[0])
        stash = ws.get(("nerf_stash", n, prec), lambda: ops.NerfStash(n, anchor.device, prec, gprec=ws.gprec)) if train else None
        density, rgb = ops.nerf_fwd(net.blob, n, prec, pts4, dirs, stash, train, n_dev)
        ctx.net, ctx.prec, ctx.ws, ctx.stash, ctx.n, ctx.n_dev = net, prec, ws, stash, n, n_dev
        ctx.generation = ws.stamp(stash) if stash is not None else None
        return density, rgb

    @staticmethod
    def backward(ctx, d_density, d_rgb):
        n, prec, net, ws = ctx.n, ctx.prec, ctx.net, ctx.ws
        ws.check(ctx.stash, ctx.generation, "background NeRF backward")
        dev = net.blob.device
        d_density = torch.zeros(n, device=dev) if d_density is None else d_density.contiguous()
        d_rgb = torch.zeros(n, 3, device=dev) if d_rgb is None else d_rgb.contiguous()
        ops.nerf_bwd(net.blob, n, prec, d_density, d_rgb, ctx.stash, ctx.n_dev)
        rec = dict(ws=ws, net=net, n=n, prec=prec, stash=ctx.stash, n_dev=ctx.n_dev, gen=ctx.generation,
                   key=(ctx.stash.zbar.data_ptr(), net.raw_grad.data_ptr(), 0 if ctx.n_dev is None else ctx.n_dev.data_ptr()))
        if ops.gemm_merge_enabled() and _PENDING_OPEN.get(dev) is not None:
            # an SDF network's backward is still to come in this pass (SdfValueGradFn.forward ran after this network's forward:
            # NeuSRenderer.render evaluates the background right in front of the compositing): its launch takes these products
            rec["for"] = _PENDING_OPEN[dev]
            _PENDING_NERF.setdefault(dev, {})[id(net)] = rec          # (one record per network: a stale one is replaced)

            def flush():
                left = _PENDING_NERF.get(dev, {}).pop(id(net), None)
                if left is rec:
                    _run_nerf_dw(left)

            torch.autograd.Variable._execution_engine.queue_callback(flush)
        else:
            _run_nerf_dw(rec)
        return None, None, None, None, None, None, None, None


_DN_ACC = {}            # (data_ptr, shape) of a normals tensor -> the compositing backward's gradient of it, for ColorFn.backward to add into
_PENDING_NERF = {}      # device -> {id(network): record of NerfFn.backward waiting for the SDF network's weight-gradient launch}
_PENDING_OPEN = {}      # device -> (id(workspace), stash stamp) of the latest SdfValueGradFn forward whose backward has not run


def _run_nerf_dw(rec):
    """the background network's weight-gradient products as a launch of their own"""
    jobs = rec["ws"].get(("nerf_jobs", rec["n"], rec["prec"]), lambda: ops.nerf_dw_jobs(rec["net"], rec["stash"], rec["n"]))
    jobs.run(n_dev=rec["n_dev"])


class OutsideAlphaFn(torch.autograd.Function):
    """alpha = 1 - exp(-softplus(density) dist), colour = sigmoid(raw) of the background samples (renderer.py:137-138): one
    launch forward, one backward, instead of ~20 element-wise kernels on [B, n + n_outside] tensors"""

    @staticmethod
    def forward(ctx, density, rgb_raw, dists):
        density, rgb_raw, dists = density.contiguous(), rgb_raw.contiguous(), dists.contiguous()
        alpha, rgb = ops.outside_alpha_fwd(density, rgb_raw, dists)
        ctx.save_for_backward(density, rgb, dists)
        ctx.set_materialize_grads(False)
        return alpha, rgb

    @staticmethod
    def backward(ctx, d_alpha, d_rgb):
        density, rgb, dists = ctx.saved_tensors
        d_density, d_raw = ops.outside_alpha_bwd(density, rgb, dists, None if d_alpha is None else d_alpha.contiguous(),
                                                 None if d_rgb is None else d_rgb.contiguous())
        return d_density, d_raw, None


class OutsideAlphaSelFn(torch.autograd.Function):
    """OutsideAlphaFn over the list of ops.outside_select: density / rgb_raw are rows of the list, the outputs the full [B, nt]
    arrays render_core reads (zeros at the samples that are not listed: their values are multiplied by 0 there)"""

    @staticmethod
    def forward(ctx, density, rgb_raw, s):
        density, rgb_raw = density.contiguous(), rgb_raw.contiguous()
        alpha, rgb = ops.outside_alpha_sel_fwd(density, rgb_raw, s)
        ctx.save_for_backward(density, rgb)
        ctx.lists = (s.dists, s.sel, s.count)
        ctx.set_materialize_grads(False)
        return alpha, rgb

    @staticmethod
    def backward(ctx, d_alpha, d_rgb):
        density, rgb = ctx.saved_tensors
        dists, sel, count = ctx.lists
        d_density, d_raw = ops.outside_alpha_sel_bwd(density, rgb, dists, sel, count,
                                                     None if d_alpha is None else d_alpha.contiguous(),
                                                     None if d_rgb is None else d_rgb.contiguous())
        return d_density, d_raw, None


class SgRenderFn(torch.autograd.Function):
    """fneus_sg_render_fwd / _bwd: the spherical-Gaussian rendering of stage 3 (inverRender.py:314-449) for the direct and the
    indirect lobes in one launch each way.  Differentiable inputs: the light SGs [M,7] and the material [n,7] (roughness,
    diffuse albedo, specular albedo); normals, view directions, visibilities and the indirect SGs are constants."""

    @staticmethod
    def forward(ctx, lgt, mat, normal, view, vis, ind, f0: float):
        t = lambda x: None if x is None else x.detach().float().contiguous()
        lgt_c, mat_c, normal, view, vis, ind = t(lgt), t(mat), t(normal), t(view), t(vis), t(ind)
        out = ops.sg_render_fwd(lgt_c, ind, vis, normal, view, mat_c, f0)
        ctx.f0 = f0
        ctx.has_ind = ind is not None
        ctx.save_for_backward(*([lgt_c, mat_c, normal, view, vis] + ([ind] if ind is not None else [])))
        return out

    @staticmethod
    def backward(ctx, d_out):
        sv = ctx.saved_tensors
        lgt, mat, normal, view, vis = sv[:5]
        ind = sv[5] if ctx.has_ind else None
        d_mat, d_lgt = ops.sg_render_bwd(lgt, ind, vis, normal, view, mat, ctx.f0, d_out.contiguous())
        return d_lgt, d_mat, None, None, None, None, None


class SgRenderHeadsFn(torch.autograd.Function):
    """SgRenderFn with the material taken straight from the two MLP heads (fneus_sg_render_heads_fwd / _bwd): brdf [n,4] = the
    BRDF decoder's sigmoid (diffuse albedo, raw roughness), cs [n,1] = net_cs's output -- the five element-wise launches that
    assembled the [n,7] table (split, 0.9 r + 0.09, expand, cat) and their five in the backward are gone.  direct_lgt: the light
    table's gradient is accumulated STRAIGHT into `lgt.grad` (the trainers' persistent, per-step cleared buffer: no zero fill,
    no `grad += temporary`) and autograd gets none for it."""

    @staticmethod
    def forward(ctx, lgt, brdf, cs, normal, view, vis, ind, f0: float, direct_lgt: bool):
        t = lambda x: None if x is None else x.detach().float().contiguous()
        lgt_c, brdf_c, cs_c, normal, view, vis, ind = t(lgt), t(brdf), t(cs), t(normal), t(view), t(vis), t(ind)
        out = ops.sg_render_heads_fwd(lgt_c, ind, vis, normal, view, brdf_c, cs_c, f0)
        ctx.f0, ctx.has_ind = f0, ind is not None
        ctx.lgt_param = lgt if direct_lgt else None
        ctx.save_for_backward(*([lgt_c, brdf_c, cs_c, normal, view, vis] + ([ind] if ind is not None else [])))
        return out

    @staticmethod
    def backward(ctx, d_out):
        sv = ctx.saved_tensors
        lgt, brdf, cs, normal, view, vis = sv[:6]
        ind = sv[6] if ctx.has_ind else None
        into = None if ctx.lgt_param is None else ctx.lgt_param.grad
        d_brdf, d_cs, d_lgt = ops.sg_render_heads_bwd(lgt, ind, vis, normal, view, brdf, cs, ctx.f0, d_out.contiguous(), into)
        return (None if into is not None else d_lgt), d_brdf, d_cs, None, None, None, None, None, None


class Stage2LossFn(torch.autograd.Function):
    """fneus_stage2_loss: the two L1 terms of a stage-2 step over the rays with a hit and their gradients (lvis.py:164-170)"""

    @staticmethod
    def forward(ctx, pre_lvis, pre_rad, gt_lvis, gt_rad, hit):
        c = lambda t: t.detach().float().contiguous()
        out, d_l, d_r = ops.stage2_loss(c(gt_lvis), c(pre_lvis), c(gt_rad), c(pre_rad),
                                        (hit.view(torch.uint8) if hit.dtype == torch.bool else hit).contiguous())
        ctx.save_for_backward(d_l, d_r)
        ctx.shapes = (pre_lvis.shape, pre_rad.shape)
        ctx.mark_non_differentiable(out)
        return out[0] + out[1], out         # (the sum: a tensor of its own that carries the gradient)

    @staticmethod
    def backward(ctx, d_loss, _d_out):
        d_l, d_r = ctx.saved_tensors
        if d_loss is ops.UNIT_LOSS_SEED:          # loss.backward(one) inside `with ops.unit_loss_grad(one)`: the factor is 1
            return d_l.reshape(ctx.shapes[0]), d_r.reshape(ctx.shapes[1]), None, None, None
        return (d_l * d_loss).reshape(ctx.shapes[0]), (d_r * d_loss).reshape(ctx.shapes[1]), None, None, None


class Stage3LossFn(torch.autograd.Function):
    """fneus_stage3_loss: the masked L1 colour term and the psnr of a stage-3 step, with the gradient of the former"""

    @staticmethod
    def forward(ctx, rgb, true_rgb, mask, hit):
        out, d_rgb = ops.stage3_loss(rgb.contiguous(), true_rgb.contiguous(), mask.reshape(-1).contiguous(),
                                     (hit.view(torch.uint8) if hit.dtype == torch.bool else hit).contiguous())
        ctx.save_for_backward(d_rgb)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out          # (the loss as its own tensor: `out` carries no gradient)

    @staticmethod
    def backward(ctx, d_loss, _d_out):
        (d_rgb,) = ctx.saved_tensors
        if d_loss is ops.UNIT_LOSS_SEED:          # loss.backward(one) inside `with ops.unit_loss_grad(one)`: the factor is 1
            return d_rgb, None, None, None
        return d_rgb * d_loss, None, None, None


class LatentKlFn(torch.autograd.Function):
    """fneus_latent_kl_fwd / _bwd: the latent-sparsity term of stage 3 over the marked points (inverRender.py:609-612)"""

    @staticmethod
    def forward(ctx, latent, point_mask, rho: float, activated: bool = False):
        """activated: `latent` is sigmoid(latent) already (the encoder's last layer applied it)"""
        latent = latent.contiguous()
        mask = None if point_mask is None else (point_mask.view(torch.uint8) if point_mask.dtype == torch.bool else point_mask).contiguous()
        stats = ops.latent_kl_fwd(latent, mask, rho, activated)
        ctx.save_for_backward(latent, stats)
        ctx.mask, ctx.rho, ctx.activated = mask, rho, activated
        return stats[33]

    @staticmethod
    def backward(ctx, d_kl):
        latent, stats = ctx.saved_tensors
        return ops.latent_kl_bwd(latent, ctx.mask, ctx.rho, stats, d_kl.reshape(1).contiguous(), ctx.activated), None, None, None


class SgCombineFn(torch.autograd.Function):
    """fneus_sg_combine_fwd / _bwd: the clamps, sums and tone mapping between the lobe sums and the rendered colour of stage 3"""

    @staticmethod
    def forward(ctx, sums, has_indir: bool):
        sums = sums.contiguous()
        ctx.save_for_backward(sums)
        ctx.has_indir = has_indir
        return ops.sg_combine_fwd(sums, has_indir)

    @staticmethod
    def backward(ctx, d_rgb):
        (sums,) = ctx.saved_tensors
        return ops.sg_combine_bwd(sums, d_rgb.contiguous(), ctx.has_indir), None


class RefHeadsFn(torch.autograd.Function):
    """Both MLPs of RefColor (fields.py:303-330) on the gathered surface samples: one forward launch, one backward
    launch, one weight-gradient GEMM launch for the two networks.  Differentiable inputs: normal, feature."""

    @staticmethod
    def forward(ctx, anchor, normal, feat, net_cd, net_vd, samples: RaySamples, prec: int, ws: _Workspace, train: bool):
        n = samples.n
        normal, feat = normal.contiguous(), feat.contiguous()
        st = ws.get(("ref_stash", n, prec), lambda: (ops.ColStash(n, anchor.device, prec, with_feat=True, gprec=ws.gprec),
                                                     ops.ColStash(n, anchor.device, prec, with_feat=True, gprec=ws.gprec))) if train else (None, None)
        diffuse, spec = ops.refcolor_fwd_both(net_cd.blob, net_vd.blob, n, prec, normal, feat, st[0], st[1], train,
                                              dirs=samples.dirs, **samples.kw())
        ctx.nets, ctx.prec, ctx.ws, ctx.st, ctx.n, ctx.samples = (net_cd, net_vd), prec, ws, st, n, samples
        ctx.gathered = _take_gathered(feat, normal)          # inputs straight from SurfaceGatherFn? -> (SDF workspace, sel, stamp)
        ctx.generation = ws.stamp(st[0]) if st[0] is not None else None
        ctx.save_for_backward(diffuse, spec, normal)
        return diffuse, spec

    @staticmethod
    def backward(ctx, d_diffuse, d_spec):
        diffuse, spec, normal = ctx.saved_tensors
        (net_cd, net_vd), n, prec, ws, st, sm = ctx.nets, ctx.n, ctx.prec, ctx.ws, ctx.st, ctx.samples
        ws.check(st[0], ctx.generation, "RefColor backward")
        z = lambda g: torch.zeros(n, 3, device=normal.device) if g is None else g.contiguous()
        d_feat2, d_normal2 = ops.refcolor_bwd_both(net_cd.blob, net_vd.blob, n, prec, z(d_diffuse), z(d_spec), diffuse, spec,
                                                   st[0], st[1], normal, dirs=sm.dirs, rays_d=sm.rays_d, m=sm.m)
        # one zeroed gradient buffer and one GEMM launch for both networks
        grad = ws.get(("ref_grad", n), lambda: torch.zeros(net_cd.n_params + net_vd.n_params, dtype=torch.float32,
                                                           device=normal.device))
        g_cd, g_vd = grad[:net_cd.n_params], grad[net_cd.n_params:]

        def build():
            jobs = ops.GemmPPJobs(grad.device, "refcolor")
            ops.color_dw_jobs(net_cd, st[0].feat, st[0], g_cd, n, into=jobs)
            ops.color_dw_jobs(net_vd, st[1].feat, st[1], g_vd, n, into=jobs)
            return jobs.finalize(st[0].tiles)

        ref = dict(ws=ws, n=n, prec=prec, build=build, nets=(net_cd, net_vd), grads=(g_cd, g_vd), st=st, gen=ctx.generation,
                   **{"for": None if ctx.gathered is None else ctx.gathered[2]},
                   key=(st[0].zbar.data_ptr(), st[1].zbar.data_ptr(), grad.data_ptr()))
        early = getattr(ctx.gathered[0], "color_grads_early", None) if ctx.gathered is not None else None
        if ctx.gathered is not None and ops.gemm_merge_enabled() and not (early is not None and early()):
            # rows gathered from the SDF network's outputs: its backward follows in this pass, and its weight-gradient launch
            # takes these products along (ColorFn.backward does the same with the colour network's)
            sdf_ws = ctx.gathered[0]
            sdf_ws.cache["pending_ref_dw"] = ref

            def flush():
                if sdf_ws.cache.get("pending_ref_dw") is ref:
                    _run_ref_dw(sdf_ws.cache.pop("pending_ref_dw"))

            torch.autograd.Variable._execution_engine.queue_callback(flush)
        else:
            with ops.on_side_stream(2):          # only Adam consumes these: off the critical path of the backward
                _run_ref_dw(ref)
        if ctx.gathered is not None:
            # the rows came from surface_gather: their per-head gradients go to the SDF backward as they are (one launch adds the
            # heads and scatters the rows) instead of sum -> SurfaceGatherFn.backward -> index_add_ (four launches)
            sdf_ws, sel, made_for = ctx.gathered
            recs = sdf_ws.cache.setdefault("surface_head_grads", {})
            recs[made_for] = (sel, d_feat2, d_normal2, made_for)
            while len(recs) > 4:                         # (records of forwards whose backward never ran: oldest first)
                recs.pop(next(iter(recs)))
            return None, None, None, None, None, None, None, None, None
        return None, d_normal2.sum(0), d_feat2.sum(0), None, None, None, None, None, None


class CompositeFn(torch.autograd.Function):
    """K5 forward / backward.  Differentiable inputs: sdf [N], normal [N,3], rgb [N,3], variance (the scalar parameter
    of SingleVarianceNetwork; inv_s = clip(exp(10 variance)) is applied inside the kernels) and, for the womask
    background model, bg_alpha [B,n+n_out], bg_color [B,n+n_out,3].
    Differentiable outputs: color [B,3], weights [B,n(+n_out)], wsum [B], wpair [B,2], eik_num [B]."""

    @staticmethod
    def forward(ctx, sdf, normal, rgb, variance, rays_o, rays_d, mid_z, dists, car: float, bg_alpha=None, bg_color=None,
                back_rgb=None):
        """back_rgb [1, 3] / [B, 3] (a constant): color + back_rgb (1 - wsum) of renderer.py:367-368, inside the kernels"""
        bga = None if bg_alpha is None else bg_alpha.contiguous()
        bgc = None if bg_color is None else bg_color.contiguous()
        var1 = variance.detach().reshape(1).contiguous()
        out = ops.composite_fwd(rays_o, rays_d, mid_z, dists, sdf.contiguous(), normal.contiguous(), rgb.contiguous(),
                                var1, car, bga, bgc, inv_s_mode=1, back_rgb=back_rgb)
        ctx.car, ctx.has_bg, ctx.var_shape, ctx.back_rgb = car, bga is not None, variance.shape, back_rgb
        ctx.var_param = variance if (variance.is_leaf and variance.requires_grad) else None
        ctx.set_materialize_grads(False)        # unused outputs (e.g. `weights`) must not be zero-filled for us
        saved = [sdf, normal, rgb, var1, rays_o, rays_d, mid_z, dists, out["min_idx"], out["sdf_mask"]]
        if ctx.has_bg:
            saved += [bga, bgc]
        ctx.save_for_backward(*saved)
        eik_num, eik_den = out["eik"][0], out["eik"][1]
        nd = (out["wmax"], out["cdf"], out["inside"], eik_den, out["min_idx"], out["sdf_mask"])
        ctx.mark_non_differentiable(*nd)
        return (out["color"], out["weights"], out["wsum"], out["wpair"], eik_num) + nd

    @staticmethod
    def backward(ctx, d_color, d_weights, d_wsum, d_wpair, d_eiknum, *unused):
        sv = ctx.saved_tensors
        sdf, normal, rgb, var1, rays_o, rays_d, mid_z, dists, min_idx, sdf_mask = sv[:10]
        bga, bgc = (sv[10], sv[11]) if ctx.has_bg else (None, None)
        B, n = mid_z.shape
        dev = mid_z.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        d_color = z(B, 3) if d_color is None else d_color.contiguous()
        d_wsum = z(B) if d_wsum is None else d_wsum.contiguous()
        d_wpair = z(B, 2) if d_wpair is None else d_wpair.contiguous()
        d_eiknum = z(B) if d_eiknum is None else d_eiknum.contiguous()
        d_weights = None if d_weights is None else d_weights.contiguous()
        d_sdf, d_normal, d_rgb, d_var, d_bga, d_bgc = ops.composite_bwd(
            rays_o, rays_d, mid_z, dists, sdf, normal, rgb, var1, ctx.car, min_idx, sdf_mask, d_color, d_wsum, d_weights,
            d_wpair, d_eiknum, bga, bgc, inv_s_mode=1, back_rgb=ctx.back_rgb)
        # the variance parameter's gradient = the sum of the per-ray terms.  With a persistent gradient buffer (the trainers' arena) and
        # the colour network's exact output-layer launch still to come in this backward pass, the sum rides in that launch's fold
        # stage and is added to variance.grad there: a reduction launch and autograd's accumulation launch less per step
        # the colour network's backward follows in this pass (its d_rgb is formed here) and differentiates the same normals: it adds its
        # gradient into this one instead of autograd summing the two in a launch of its own (ColorFn.backward takes the record; whatever
        # is left when the pass ends is dropped -- the tensor itself goes to autograd as ever)
        if ops.DNORMAL_ACC and d_normal is not None:
            key = (normal.data_ptr(), tuple(normal.shape))
            _DN_ACC[key] = d_normal
            torch.autograd.Variable._execution_engine.queue_callback(lambda: _DN_ACC.pop(key, None))
        var = ctx.var_param
        if (var is not None and var.grad is not None and var.grad.is_contiguous() and var.grad.dtype == torch.float32 and var.numel() == 1
                and ops.DEFAULT_FOLD_RIDER and ops.PROFILE is None):
            ops.offer_fold_rider(d_var, var.grad)
            d_var_out = None
        else:
            d_var_out = d_var.sum().reshape(ctx.var_shape)
        return d_sdf, d_normal, d_rgb, d_var_out, None, None, None, None, None, d_bga, d_bgc, None


# The latest surface_gather's outputs, held WEAKLY and recognised by identity (an address can be handed to an unrelated tensor of
# the same shape by the caching allocator once the rows are freed): (ref(feat_sel), ref(normal_sel), SDF workspace, sel, stamp of
# the SDF forward the rows came from).  One slot, taken by the first consumer that asks for exactly these tensors.
_GATHERED = []


def _take_gathered(feat, normal):
    if not _GATHERED:
        return None
    rf, rn, sdf_ws, sel, made_for = _GATHERED[0]
    if rf() is feat and rn() is normal:
        _GATHERED.clear()
        return sdf_ws, sel, made_for
    return None


class SurfaceGatherFn(torch.autograd.Function):
    """Rows of (feature, normal, depth) at the two samples bracketing the first SDF sign change of every ray
    (renderer.py:290-293, 316-327), one launch.  Its backward hands the 2B gradient rows to the SDF network's backward
    through the workspace (see SdfValueGradFn.backward) and returns no dense gradient."""

    @staticmethod
    def forward(ctx, feat, normal, mid_z, min_idx, sdf_mask, sdf_ws: _Workspace, warm=None):
        sel, t_sel, feat_sel, normal_sel = ops.surface_gather(min_idx, sdf_mask, mid_z.contiguous(), feat.contiguous(),
                                                              normal.contiguous(), warm=warm)
        ctx.sdf_ws = sdf_ws
        # a consumer that produces per-head gradients of exactly these rows (RefHeadsFn) may hand them to the SDF backward
        # directly (fneus_surface_scatter) instead of summing them and going through this function's backward
        _GATHERED.clear()                    # (one slot: the consumer looks it up right behind this call, in the same step)
        _GATHERED.append((weakref.ref(feat_sel), weakref.ref(normal_sel), sdf_ws, sel,
                          (id(sdf_ws), getattr(sdf_ws, "generation", None))))
        ctx.save_for_backward(sel)
        ctx.mark_non_differentiable(t_sel, sel)
        ctx.set_materialize_grads(False)         # no zero-filled cotangents for the two index outputs
        return feat_sel, normal_sel, t_sel, sel

    @staticmethod
    def backward(ctx, d_feat_sel, d_normal_sel, _t, _s):
        (sel,) = ctx.saved_tensors
        ctx.sdf_ws.cache["surface_grads"] = (sel, None if d_feat_sel is None else d_feat_sel.contiguous(),
                                             None if d_normal_sel is None else d_normal_sel.contiguous())
        return None, None, None, None, None, None, None


class Stage1LossFn(torch.autograd.Function):
    """RefColor shading + two-sample blend + the four training losses, with the gradients of the total loss computed in
    the same launch (csrc/loss_kernels.hip).  Differentiable inputs: color [B,3], wsum [B], eik_num [B], wpair [B,2],
    diffuse [2B,3], spec [2B,3] (column 0).  Returns loss (scalar), losses [8], surface / specular / diffuse colour."""

    @staticmethod
    def forward(ctx, color, wsum, eik_num, wpair, diffuse, spec, eik_den, true_rgb, mask_in, sdf_mask, igr_weight,
                mask_weight, surface_weight, reduce_norms=None, warm=None):
        mask_flat = mask_in.contiguous().reshape(-1)
        norms = None
        if reduce_norms is not None:       # data parallel: normalisers of the global batch
            norms = reduce_norms(ops.stage1_norms(mask_flat, sdf_mask, eik_den.contiguous(), mask_weight))
        o = ops.stage1_loss(color.contiguous(), true_rgb.contiguous(), mask_flat, wsum.contiguous(),
                            eik_num.contiguous(), eik_den.contiguous(), diffuse.contiguous(), spec.contiguous(),
                            wpair.contiguous(), sdf_mask, igr_weight, mask_weight, surface_weight, norms=norms, warm=warm)
        ctx.save_for_backward(o["d_color"], o["d_wsum"], o["d_eiknum"], o["d_wpair"], o["d_diffuse"], o["d_spec"])
        aux = (o["losses"], o["surface_color"], o["specular_color"], o["diffuse_color"])
        ctx.mark_non_differentiable(*aux)
        ctx.set_materialize_grads(False)         # no zero-filled cotangents for the four report-only outputs
        return (o["loss"],) + aux          # (slot 8 of the kernel's losses: the total as a tensor of its own, no copy launch)

    @staticmethod
    def backward(ctx, g, *unused):
        if g is None:
            return (None,) * 15
        seed = ops.UNIT_LOSS_SEED            # the trainers' `loss.backward(one)`: the cotangent IS their persistent constant 1
        if seed is not None and (g is seed or (g.data_ptr() == seed.data_ptr() and g.shape == seed.shape)):
            return tuple(ctx.saved_tensors) + (None,) * 9
        grads = torch._foreach_mul(list(ctx.saved_tensors), g)
        return tuple(grads) + (None,) * 9
