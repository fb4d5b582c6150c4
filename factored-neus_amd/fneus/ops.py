"""Thin torch-tensor wrappers over the C ABI (device memory + stream plumbing only; no maths here)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, netdesc
from ._lib import lib, check

PREC_FAST, PREC_PARITY = 1, 3
PREC_H16 = 2          # ONE fp16 product per multiplication: the stage-3 visibility network only (fneus_lvis_visibility)

# Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline figures).
# PROFILE = None disables it (default; the hot path then records nothing and never synchronises).
PROFILE = None


def profile_begin():
    global PROFILE
    PROFILE = []


def profile_end():
    """-> {kernel name: (launches, total milliseconds)}"""
    global PROFILE
    rec, PROFILE = PROFILE, None
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in rec:
        n, t = out.get(name, (0, 0.0))
        out[name] = (n + 1, t + e0.elapsed_time(e1))
    return out


# ------------------------------------------------------------------------------------------------------------
# Side stream: work that nothing on the critical path of a step waits for (the packs of the networks the sampler does
# not use; the weight-gradient GEMMs + fold backward of the RefColor heads and the colour network, which only Adam
# consumes) is issued on a second HIP stream so that it fills the machine beside latency-bound kernels.  Opt-in per
# step: the trainer opens a window with overlap_begin() and closes it (joining the stream) with overlap_end(); outside
# such a window every op stays on the caller's stream.  FNEUS_OVERLAP is a bit mask: 1 packs, 2 RefColor dW, 4 colour dW.
# MEASURED (MI355X, replayed step, 512 x 128): 3.512 ms off, 3.548 / 3.546 / 3.564 ms with bit 1 / 2 / 4 alone, 3.522 ms
# with all three -- no gain (the forked branches of the captured graph do not run beside the main one often enough to pay
# for the extra dependencies), so the default is OFF; kept as a switch for multi-queue experiments on other shapes.
# ------------------------------------------------------------------------------------------------------------
import os as _os
import os
OVERLAP_MASK = int(_os.environ.get("FNEUS_OVERLAP", "0"))
_overlap = {"on": False, "stream": None, "used": False}


def overlap_begin(device):
    if OVERLAP_MASK == 0 or PROFILE is not None:
        return
    if _overlap["stream"] is None:
        _overlap["stream"] = torch.cuda.Stream(device=device)
    _overlap["on"], _overlap["used"] = True, False


def overlap_join():
    """make the current stream wait for everything issued on the side stream so far"""
    if _overlap["used"]:
        torch.cuda.current_stream().wait_stream(_overlap["stream"])
        _overlap["used"] = False


def overlap_end():
    overlap_join()
    _overlap["on"] = False


class on_side_stream:
    """with on_side_stream(bit): ...   -- the body runs on the side stream, ordered after everything issued so far on the
    current one; a no-op outside an overlap window or when the bit is not enabled"""

    def __init__(self, bit: int):
        self.active = _overlap["on"] and bool(OVERLAP_MASK & bit)
        self.ctx = None

    def __enter__(self):
        if self.active:
            side = _overlap["stream"]
            side.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
            _overlap["used"] = True
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _launch(name, fn, *args):
    if PROFILE is None:
        check(fn(*args), name)
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), name)
    e1.record()
    PROFILE.append((name, e0, e1))


# The fused loss kernel hands out the gradients of the TOTAL loss; autograd multiplies them by the cotangent of the loss -- a launch
# per step for a factor that is the constant 1 when the trainer calls loss.backward(one).  Inside `with unit_loss_grad(one):` a
# loss node whose cotangent IS that tensor (identity, not value: a loss-scaling wrapper or a hook that multiplies the loss hands
# autograd another tensor and gets the multiply) passes the saved gradients on as they are.
UNIT_LOSS_SEED = None


class unit_loss_grad:
    def __init__(self, seed):
        self.seed = seed

    def __enter__(self):
        global UNIT_LOSS_SEED
        self.prev, UNIT_LOSS_SEED = UNIT_LOSS_SEED, self.seed
        return self

    def __exit__(self, *exc):
        global UNIT_LOSS_SEED
        UNIT_LOSS_SEED = self.prev
        return False


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_f32(t: torch.Tensor, name: str):
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 CUDA tensor")


MAX_WARM_RANGES = 12


class WarmRanges(C.Structure):
    """FneusWarmRanges (include/fneus.h): device ranges that the extra workgroups of ONE surface_gather / stage1_loss launch read
    into L2 for the launch behind it.  Built by the owner of the buffers (models/fields.py RefColor) and handed to the call:
    the library keeps nothing of it."""
    _fields_ = [("n", C.c_int), ("ptr", C.c_void_p * MAX_WARM_RANGES), ("bytes", C.c_long * MAX_WARM_RANGES)]


def warm_ranges(ranges):
    """[(ptr, bytes), ...] -> WarmRanges, or None (no ranges, or FNEUS_L2_WARM=0)"""
    if not ranges or _os.environ.get("FNEUS_L2_WARM", "1") == "0":
        return None
    if len(ranges) > MAX_WARM_RANGES:
        raise ValueError(f"at most {MAX_WARM_RANGES} warm-up ranges per launch")
    w = WarmRanges()
    w.n = len(ranges)
    for i, (p, b) in enumerate(ranges):
        w.ptr[i] = int(p)
        w.bytes[i] = int(b)
    return w


def _warm_arg(w):
    return None if w is None else C.byref(w)


def fragment_ranges(net, reverse: bool):
    """(ptr, bytes) of a packed network's forward (or reverse) weight fragments, hi + lo, layer by layer"""
    out = []
    base = net.blob.data_ptr()
    for l in range(net.layout.n_layers):
        ksf, ntf, ksr, ntr = [int(v) for v in net.layout.geom[l]]
        fwd_hi, fwd_lo, rev_hi, rev_lo, _bias = [int(v) for v in net.layout.off[l]]
        if reverse:
            if ksr * ntr > 0:
                out.append((base + rev_hi, 2 * ksr * ntr * 1024))
        else:
            out.append((base + fwd_hi, 2 * ksf * ntf * 1024))
    return out


# Networks whose pack() was called inside `with batched_refresh():` are folded and packed by ONE fneus_refresh_multi call at the
# end of the block (two launches for all of them instead of one or two each: a training step re-packs 4-5 networks at its
# start).  FNEUS_PACK_BATCH=0 keeps the per-network launches.
_pack_batch = None
PACK_BATCH = _os.environ.get("FNEUS_PACK_BATCH", "1") != "0"


class batched_refresh:
    def __enter__(self):
        global _pack_batch
        self.outer = _pack_batch
        if PACK_BATCH and _pack_batch is None and PROFILE is None:
            _pack_batch = []
        return self

    def __exit__(self, *exc):
        global _pack_batch
        if self.outer is None and _pack_batch is not None:
            nets, _pack_batch = _pack_batch, None
            if exc[0] is None and nets:
                if len(nets) == 1:
                    nets[0].pack()
                else:
                    tasks = (_lib.FneusPackTask * len(nets))()
                    for t, n in zip(tasks, nets):
                        wn = n.desc.get("weight_norm", True)
                        t.jobs, t.n_jobs, t.n_units, t.maps = n.jobs.data_ptr(), n.n_jobs, n.units, n.maps.data_ptr()
                        t.params, t.blob = n.raw.data_ptr(), n.blob.data_ptr()
                        t.rowscale = n.rowscale.data_ptr() if n.rowscale.numel() else None
                        t.invnorm = n.invnorm.data_ptr() if n.invnorm.numel() else None
                        t.rows, t.n_rows = (n.rows.data_ptr(), n.n_rows) if wn and n.n_rows else (None, 0)
                    check(lib.fneus_refresh_multi(tasks, len(nets), _stream()), "fneus_refresh_multi")
        return False


# The same for fneus_wn_backward: inside `with batched_wn_backward():` (around loss.backward() of a single-GPU step: only the
# optimiser reads the raw gradients) the calls are collected and launched once at the end of the block.
_wn_batch = None


MAX_WN_TASKS = 8         # csrc/pack.hip kMaxWnTasks: task table by value in the kernel arguments


def _launch_wn_calls(calls):
    global _wn_batch
    if not calls:
        return
    # one task per (network, d_eff): a network collected twice would have two tasks racing on its raw gradient inside the launch
    seen, uniq = set(), []
    for n, d_eff in calls:
        key = (id(n), d_eff.data_ptr())
        if key not in seen:
            seen.add(key)
            uniq.append((n, d_eff))
    calls = uniq
    if len({id(n) for n, _ in calls}) != len(calls) or len(calls) > MAX_WN_TASKS:
        # the same network with two different d_eff buffers, or more tasks than one launch takes: one launch each, in order
        saved, _wn_batch = _wn_batch, None
        try:
            for n, d_eff in calls:
                n.wn_backward(d_eff)
        finally:
            _wn_batch = saved
        return
    if len(calls) == 1:
        saved, _wn_batch = _wn_batch, None
        try:
            calls[0][0].wn_backward(calls[0][1])
        finally:
            _wn_batch = saved
        return
    tasks = (_lib.FneusWnTask * len(calls))()
    for t, (n, d_eff) in zip(tasks, calls):
        t.rows, t.n_rows = (n.rows.data_ptr() if n.n_rows else None), n.n_rows
        t.bias_segs, t.n_segs = (n.bias_segs.data_ptr() if n.bias_segs.numel() else None), int(n.bias_segs.shape[0])
        t.raw, t.d_eff, t.d_raw = n.raw.data_ptr(), d_eff.data_ptr(), n.raw_grad.data_ptr()
        t.rowscale = n.rowscale.data_ptr() if n.rowscale.numel() else None
        t.invnorm = n.invnorm.data_ptr() if n.invnorm.numel() else None
    check(lib.fneus_wn_backward_multi(tasks, len(calls), _stream()), "fneus_wn_backward_multi")


def flush_wn_batch():
    """launch what `batched_wn_backward` has collected so far (data parallel: before the early part of the gradient arena is
    exchanged); the block goes on collecting"""
    global _wn_batch
    if _wn_batch:
        calls, _wn_batch = _wn_batch, []
        _launch_wn_calls(calls)


class batched_wn_backward:
    def __enter__(self):
        global _wn_batch
        self.outer = _wn_batch
        if PACK_BATCH and _wn_batch is None and PROFILE is None:
            _wn_batch = []
        return self

    def __exit__(self, *exc):
        global _wn_batch
        if self.outer is None and _wn_batch is not None:
            calls, _wn_batch = _wn_batch, None
            # ALSO when the block exits on an exception: the weight-gradient GEMMs that ran have accumulated into the d_eff
            # buffers, which fneus_wn_backward consumes AND clears -- dropping the calls would leave them non-zero and the next
            # step would silently add stale gradients (the unbatched path clears each buffer right behind its GEMM)
            _launch_wn_calls(calls)
        return False


class PackedNet:
    """Device-side packed weights of one MLP ('sdf' or 'color'), its pack-job tables and the flat RAW parameter /
    gradient buffers (bias, weight_g, weight_v per layer, state_dict order).  The weight-norm fold runs inside the
    packer and its backward in fneus_wn_backward, so a step needs 2 + 2 small launches per network instead of ~150
    torch element-wise kernels."""

    def __init__(self, kind: str, device, raw_grad: Optional[torch.Tensor] = None):
        builders = {"sdf": netdesc.build_sdf_jobs, "color": netdesc.build_color_jobs,
                    "refcd": netdesc.build_refcd_jobs, "refvd": netdesc.build_refvd_jobs,
                    "nerf": netdesc.build_nerf_jobs, "lvis": netdesc.build_lvis_jobs}
        desc = builders[kind]()
        self.kind, self.desc, self.device = kind, desc, device
        self.layout = desc["layout"]
        self.n_params = desc["n_params"]            # effective (W, b) layout: gradient buffer of the dW GEMM
        self.n_raw = desc["n_raw"]
        self.jobs = torch.from_numpy(desc["jobs"].view(np.uint8).copy()).to(device)
        self.maps = torch.from_numpy(desc["maps"]).to(device)
        self.rows = torch.from_numpy(desc["rows"].view(np.uint8).copy()).to(device)
        self.bias_segs = torch.from_numpy(desc["bias_segs"].copy()).to(device)
        self.n_rows = len(desc["rows"])
        self.n_jobs = len(desc["jobs"])
        self.units = desc["units"]
        self.blob = torch.zeros(self.layout.total, dtype=torch.uint8, device=device)
        self.rowscale = torch.zeros(self.n_rows, dtype=torch.float32, device=device)
        self.invnorm = torch.zeros(self.n_rows, dtype=torch.float32, device=device)
        self.raw = torch.zeros(self.n_raw, dtype=torch.float32, device=device)
        if raw_grad is None:
            raw_grad = torch.zeros(self.n_raw, dtype=torch.float32, device=device)
        elif raw_grad.numel() != self.n_raw or raw_grad.dtype != torch.float32 or not raw_grad.is_contiguous():
            raise ValueError(f"raw_grad must be a contiguous fp32 tensor of {self.n_raw} values")
        self.raw_grad = raw_grad      # may be a slice of a model-wide gradient arena (fneus/parallel.py GradArena)

    # ---- raw parameter views (what the nn.Parameters of the module alias) ----
    def raw_views(self, buf):
        d = self.desc
        out = []
        for l, (i, o) in enumerate(zip(d["ins"], d["outs"])):
            if d.get("weight_norm", True):
                out.append({"bias": buf[d["offB"][l]: d["offB"][l] + o],
                            "weight_g": buf[d["offG"][l]: d["offG"][l] + o].view(o, 1),
                            "weight_v": buf[d["offV"][l]: d["offV"][l] + o * i].view(o, i)})
            else:       # plain nn.Linear
                out.append({"bias": buf[d["offB"][l]: d["offB"][l] + o],
                            "weight": buf[d["offV"][l]: d["offV"][l] + o * i].view(o, i)})
        return out

    def load_state_dict(self, sd, prefix="lin"):
        """raw parameters from a module-style dict (lin{l}.bias / weight_g / weight_v, or .weight for plain layers)"""
        for l, view in enumerate(self.raw_views(self.raw)):
            for name, dst in view.items():
                dst.copy_(torch.as_tensor(sd[f"{prefix}{l}.{name}"]).reshape(dst.shape).to(dst.device))
        return self

    def set_raw_from_effective(self, Ws, bs):
        """test helper: load effective weights as (g = ||W||, v = W)"""
        for view, W, b in zip(self.raw_views(self.raw), Ws, bs):
            view["bias"].copy_(b)
            if "weight" in view:
                view["weight"].copy_(W)
            else:
                view["weight_v"].copy_(W)
                view["weight_g"].copy_(W.norm(dim=1, keepdim=True))

    def split_flat(self, flat: torch.Tensor):
        Ws, bs = [], []
        for l, (i, o) in enumerate(zip(self.desc["ins"], self.desc["outs"])):
            Ws.append(flat[self.desc["offW"][l]: self.desc["offW"][l] + i * o].view(o, i))
            bs.append(flat[self.desc["offb"][l]: self.desc["offb"][l] + o])
        return Ws, bs

    def pack(self):
        """fold weight-norm + pack the current raw parameters (once per optimiser step)"""
        if _pack_batch is not None:                   # inside `with batched_refresh():` -- launched with the others at its end
            if all(n is not self for n in _pack_batch):
                _pack_batch.append(self)
            return self.blob
        if self.desc.get("weight_norm", True):        # plain Linear networks have nothing to fold
            _launch("fneus_rowscale", lib.fneus_rowscale, _ptr(self.rows), self.n_rows, _ptr(self.raw), _ptr(self.rowscale),
                    _ptr(self.invnorm), _stream())
        _launch("fneus_pack", lib.fneus_pack, _ptr(self.jobs), self.n_jobs, self.units, _ptr(self.maps), _ptr(self.raw),
                _ptr(self.rowscale), _ptr(self.blob), _stream())
        return self.blob

    def wn_backward(self, d_eff: torch.Tensor):
        """accumulate raw-parameter gradients from the effective-parameter gradients"""
        if _wn_batch is not None:                     # inside `with batched_wn_backward():` -- one launch for all at its end
            _wn_batch.append((self, d_eff))
            return
        _launch("fneus_wn_backward", lib.fneus_wn_backward, _ptr(self.rows), self.n_rows, _ptr(self.bias_segs),
                int(self.bias_segs.shape[0]), _ptr(self.raw), _ptr(self.rowscale), _ptr(self.invnorm), _ptr(d_eff),
                _ptr(self.raw_grad), _stream())


# "gradient precision" of the backward stash: 1 = bf16 hi planes only (the weight-gradient GEMM multiplies bf16 operands,
# fp32 accumulation), 3 = hi + lo planes (fp32-accurate weight gradients, twice the stash traffic).  The forward outputs
# (sdf, feature, normal, colours) do not depend on it.  FNEUS_GPREC overrides the default.
DEFAULT_GPREC = int(_os.environ.get("FNEUS_GPREC", "2"))


def _gprec(prec: int, gprec: Optional[int], mixed: bool = False) -> int:
    """gradient precision of a stash: 1 = hi planes, 3 = hi + lo planes, 2 (round 5) = hi planes everywhere but for the ONE product
    whose bf16 rounding exceeds the exact mode's bounds -- the colour network's output layer (3 x 256: zout^T u_3; see
    tools/experiments/r05/gprec_tensors.py) -- which keeps its two lo planes.  Only the colour network's stash knows mode 2
    (mixed=True); for every other stash it is mode 1."""
    g = DEFAULT_GPREC if gprec is None else gprec
    if prec == 1:
        return 1
    return g if (g != 2 or mixed) else 1


class SdfStash:
    """Activation stash written by sdf_fwd_grad (include/fneus.h FneusSdfStash): fragment planes (fneus/pp.py) for
    h_l, a_l, the feature vector and the positional encoding, the lane-private sigma' plane."""

    def __init__(self, n: int, device, prec: int, train: bool, gprec: Optional[int] = None):
        self.n, self.prec, self.gprec = n, prec, _gprec(prec, gprec)
        bf = torch.bfloat16
        P = 2 if self.gprec == 3 else 1
        self.tiles = (n + 31) // 32                  # tiles that hold samples (what the GEMM sums over)
        T = 2 * ((n + 63) // 64)                     # tiles allocated (csrc/fneus_pp.h pp_tiles)
        self.pe = torch.zeros((P, T, 4, 64, 8), dtype=bf, device=device) if train else None      # fragment 3 stays zero
        self.h = torch.empty((P, 8, T, 16, 64, 8), dtype=bf, device=device) if train else None
        self.a = torch.empty((P, 8, T, 16, 64, 8), dtype=bf, device=device) if train else None
        # the feature vector: the colour network's INPUT (hi + lo planes in parity mode, whatever the gradient precision: round 6) and,
        # as its first P planes, an operand of the colour network's weight-gradient products
        self.feat = torch.empty((2 if prec == PREC_PARITY else 1, T, 16, 64, 8), dtype=bf, device=device) if train else None
        self.ps = torch.empty((T, 8, 16, 64, 8), dtype=torch.int16, device=device)      # sigma' as u16 fixed point
        self.qs = torch.empty((T, 2, 64, 16), dtype=torch.float32, device=device)       # q_skip scratch of the reverse sweep
        s = _lib.FneusSdfStash()
        s.ps = self.ps.data_ptr()
        s.qs = self.qs.data_ptr()
        for name, t in (("pe", self.pe), ("h", self.h), ("a", self.a), ("feat", self.feat)):
            if t is None:
                continue
            setattr(s, name + "_hi", t[0].data_ptr())
            setattr(s, name + "_lo", t[1].data_ptr() if t.shape[0] == 2 else None)
        self.c = s

    def plane(self, t, slot=None):
        """fp32 value [n, 16 F] of a fragment-plane tensor (hi + lo), optionally of one layer slot"""
        from . import pp
        return pp.value(t if slot is None else t[:, slot], self.n)

    def sigma(self, slot):
        """sigma'(z_slot) [n, 256] decoded from the fixed-point plane (lane-linear fragments)"""
        u = self.ps[:, slot].to(torch.int32) & 0xFFFF                       # [T, 16, 64, 8]
        T = u.shape[0]
        lane = torch.arange(64, device=u.device)
        r, h = lane & 31, lane >> 5
        out = torch.empty(T, 32, 256, dtype=torch.float32, device=u.device)
        for ks in range(16):
            for j in range(8):
                f = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)
                out[:, r, f] = u[:, ks, :, j].float() / 65535.0
        return out.reshape(T * 32, 256)[:self.n]


def sdf_fwd(blob, n_pts: int, prec: int, pts=None, rays_o=None, rays_d=None, t=None, m: int = 1, out=None, ray_mask=None,
            fill: float = 1.0):
    """ray_mask [n_pts / m] uint8 / bool (ray form, m a multiple of 128, >= 32 768 samples): only the marked rays are evaluated, the
    samples of the others get `fill` (fneus_sdf_fwd_rays)"""
    dev = blob.device
    if out is None:
        out = torch.empty(n_pts, dtype=torch.float32, device=dev)
    if ray_mask is not None and pts is None and m % 128 == 0 and n_pts >= 32768 and os.environ.get("FNEUS_K1_RAY_MASK", "1") != "0":
        mask = ray_mask.view(torch.uint8) if ray_mask.dtype == torch.bool else ray_mask
        work = torch.empty(n_pts // 128 + 1, dtype=torch.int32, device=dev)
        _launch("fneus_sdf_fwd_rays", lib.fneus_sdf_fwd_rays, _ptr(blob), _ptr(rays_o), _ptr(rays_d), _ptr(t), m, n_pts,
                _ptr(mask.contiguous()), float(fill), _ptr(work), _ptr(out), prec, _stream())
        return out
    _launch("fneus_sdf_fwd", lib.fneus_sdf_fwd, _ptr(blob), _ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(t), m, n_pts, _ptr(out), prec,
                            _stream())
    return out


FEAT_PLANES = _os.environ.get("FNEUS_FEAT_PLANES", "1") != "0"


def feat_planes_ok(n_pts: int, prec: int, train: bool) -> bool:
    """may a K2 launch leave the fp32 feature rows out (the consumers read the stash's feature planes)?  The conditions of the two
    kernels that implement it (csrc/sdf_kernels.hip fneus_sdf_fwd_grad, color_kernels.hip fneus_color_fwd): training launches of
    >= 1024 sample tiles on the two-launch K2 and the two-pass colour forward"""
    return (FEAT_PLANES and train and (n_pts + 31) // 32 >= 1024 and prec in (PREC_PARITY, PREC_FAST)
            and _os.environ.get("FNEUS_K2_P2", "1") == "1" and _os.environ.get("FNEUS_COL_P2", "1") != "0")


def sdf_fwd_grad(blob, n_pts: int, prec: int, stash: SdfStash, train: bool, pts=None, rays_o=None, rays_d=None,
                 t=None, m: int = 1, feat_rows: bool = True):
    """feat_rows False (only where feat_planes_ok): the [n, 256] tensor returned for `feat` is a PLACEHOLDER that no kernel wrote --
    the features are the stash's hi + lo planes (its `planes_of` attribute names the stash)"""
    dev = blob.device
    sdf = torch.empty(n_pts, dtype=torch.float32, device=dev)
    feat = torch.empty(n_pts, 256, dtype=torch.float32, device=dev)
    normal = torch.empty(n_pts, 3, dtype=torch.float32, device=dev)
    _launch("fneus_sdf_fwd_grad", lib.fneus_sdf_fwd_grad, _ptr(blob), _ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(t), m, n_pts,
                                 C.byref(stash.c), _ptr(sdf), _ptr(feat) if feat_rows else None, _ptr(normal), prec, int(train), _stream())
    if not feat_rows:
        feat.planes_of = stash
    return sdf, feat, normal


# ------------------------------------------------------------------------------------------------------------
# backward-side buffers and the weight-gradient GEMM
# ------------------------------------------------------------------------------------------------------------
class SdfBwdBufs:
    """work buffers of sdf_bwd (include/fneus.h FneusSdfBwdBufs): fragment planes + the coupling scratch"""

    def __init__(self, n: int, device, prec: int, gprec: Optional[int] = None):
        bf = torch.bfloat16
        self.n, self.gprec = n, _gprec(prec, gprec)
        P = 2 if self.gprec == 3 else 1
        self.tiles = (n + 31) // 32
        T = 2 * ((n + 63) // 64)
        self.qbar = torch.zeros((P, T, 4, 64, 8), dtype=bf, device=device)          # fragment 3 stays zero
        self.adj = torch.zeros((P, 8, T, 16, 64, 8), dtype=bf, device=device)
        self.zbar = torch.zeros((P, 9, T, 16, 64, 8), dtype=bf, device=device)
        self.zsdf = torch.zeros((P, T, 2, 64, 8), dtype=bf, device=device)
        self.cs = torch.empty((P, T, 8, 16, 64, 8), dtype=bf, device=device)
        # the implicit "ones" row of the sdf output (a_8 = e_0): one constant block, feature 0 = 1 for every sample
        from . import pp
        ones = torch.zeros(32, 32, device=device)
        ones[:, 0] = 1.0
        self.ones = pp.pack(ones, 2, P)                                                # [P, 1, 2, 64, 8]
        s = _lib.FneusSdfBwdBufs()
        for name, t in (("qbar", self.qbar), ("adj", self.adj), ("zbar", self.zbar), ("zsdf", self.zsdf), ("c", self.cs)):
            setattr(s, name + "_hi", t[0].data_ptr())
            setattr(s, name + "_lo", t[1].data_ptr() if P == 2 else None)
        self.c = s


class ColStash:
    """stash of a colour-shaped MLP (include/fneus.h FneusColStash): fragment planes + the lane-private ReLU masks"""

    def __init__(self, n: int, device, prec: int, with_feat: bool = False, gprec: Optional[int] = None):
        bf = torch.bfloat16
        self.n, self.gprec = n, _gprec(prec, gprec, mixed=not with_feat)
        P = 2 if self.gprec == 3 else 1
        self.tiles = (n + 31) // 32
        T = 2 * ((n + 63) // 64)
        self.feat = torch.empty((P, T, 16, 64, 8), dtype=bf, device=device) if with_feat else None
        self.side = torch.zeros((P, T, 4, 64, 8), dtype=bf, device=device)           # fragment 3 stays zero
        self.u = torch.empty((P, 4, T, 16, 64, 8), dtype=bf, device=device)
        self.zbar = torch.zeros((P, 4, T, 16, 64, 8), dtype=bf, device=device)
        self.zout = torch.zeros((P, T, 2, 64, 8), dtype=bf, device=device)
        self.mask = torch.zeros(T * 4 * 64 * 4, dtype=torch.int32, device=device)
        s = _lib.FneusColStash()
        s.mask = self.mask.data_ptr()
        for name, t in (("side", self.side), ("u", self.u), ("zbar", self.zbar), ("zout", self.zout), ("feat", self.feat)):
            if t is None:
                continue
            setattr(s, name + "_hi", t[0].data_ptr())
            setattr(s, name + "_lo", t[1].data_ptr() if P == 2 else None)
        self.u3_lo = self.zout_lo = None
        if self.gprec == 2:
            # the output layer's two operands keep their lo planes: slot 3 of u and zout.  The kernels address u_lo like u_hi
            # ([4][tiles] blocks) and, with side_lo NULL, touch slot 3 alone (include/fneus.h): the pointer is slot 3's plane minus
            # three slots that do not exist
            self.u3_lo = torch.empty((T, 16, 64, 8), dtype=bf, device=device)
            self.zout_lo = torch.zeros((T, 2, 64, 8), dtype=bf, device=device)
            s.u_lo = self.u3_lo.data_ptr() - 3 * T * 16 * 1024
            s.zout_lo = self.zout_lo.data_ptr()
        self.c = s


def sdf_bwd(blob, n_pts, prec, stash: SdfStash, bufs: SdfBwdBufs, d_sdf, d_feat, d_normal, pts=None, rays_o=None,
            rays_d=None, t=None, m: int = 1):
    """d_feat None (only where dfeat_plane_ok): the feature cotangent is in bufs.zbar[0, 8] already, as bf16 fragments"""
    for x, nm in ((d_sdf, "d_sdf"), (d_feat, "d_feat"), (d_normal, "d_normal")):
        if x is None and nm == "d_feat":
            continue
        _chk_f32(x, nm)
    _launch("fneus_sdf_bwd", lib.fneus_sdf_bwd, _ptr(blob), _ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(t), m, n_pts, C.byref(stash.c),
                            C.byref(bufs.c), _ptr(d_sdf), _ptr(d_feat), _ptr(d_normal), prec, _stream())


HEAD_COLOR, HEAD_REF_DIFFUSE, HEAD_REF_SPECULAR = 0, 1, 2     # csrc/color_kernels.hip VAR_*


def color_fwd(blob, n_pts, prec, normal, feat, stash: Optional[ColStash], train: bool, pts=None, rays_o=None,
              rays_d=None, t=None, m: int = 1, dirs=None, head: int = HEAD_COLOR):
    """colour network (head 0) or one of the two RefColor MLPs (head 1: diffuse rgb, head 2: specular in column 0)"""
    _chk_f32(normal, "normal")
    _chk_f32(feat, "feat")
    rgb = torch.empty(n_pts, 3, dtype=torch.float32, device=blob.device)
    src = getattr(feat, "planes_of", None)          # a placeholder of sdf_fwd_grad(feat_rows=False): the features are that stash's planes
    if src is not None:
        if head != HEAD_COLOR or stash is None:
            raise ValueError("feature planes instead of rows: the colour network's training forward only")
        if prec == PREC_PARITY and src.feat.shape[0] < 2:
            raise ValueError("feature planes without a lo plane (an SDF network in bf16 mode) cannot feed a colour network in parity mode")
        stash.c.feat_hi = src.feat[0].data_ptr()
        stash.c.feat_lo = src.feat[1].data_ptr() if src.feat.shape[0] == 2 else None
    sp = C.byref(stash.c) if stash is not None else None
    if head == HEAD_COLOR:
        _launch("fneus_color_fwd", lib.fneus_color_fwd, _ptr(blob), _ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(t), m, n_pts,
                _ptr(dirs), _ptr(normal), None if src is not None else _ptr(feat), sp, _ptr(rgb), prec, int(train), _stream())
    else:
        _launch("fneus_refcolor_fwd", lib.fneus_refcolor_fwd, _ptr(blob), head, _ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(t),
                m, n_pts, _ptr(dirs), _ptr(normal), _ptr(feat), sp, _ptr(rgb), prec, int(train), _stream())
    return rgb


DFEAT_PLANE = _os.environ.get("FNEUS_DFEAT_PLANE", "1") != "0"


def dfeat_plane_ok(n_pts: int, prec: int, gprec: int) -> bool:
    """may the colour network's backward hand its feature cotangent to the SDF network's backward as the bf16 fragments of that
    kernel's seed plane (slot 8 of SdfBwdBufs.zbar) instead of fp32 rows [n, 256]?  The conditions under which BOTH kernels run their
    chains on bf16 cotangents (csrc/color_r8_kernels.hip color_bwd_r8, csrc/sdf_r8_kernels.hip sdf_bwd_r8): parity arithmetic, bf16
    gradient planes, launches of >= 1024 sample tiles within the 32-bit plane offsets, none of the switches that select another kernel"""
    env = _os.environ.get
    tiles_pp = 2 * ((n_pts + 63) // 64)
    return (DFEAT_PLANE and prec == PREC_PARITY and gprec in (1, 2) and (n_pts + 31) // 32 >= 1024 and tiles_pp * 9 * 16384 < 2 ** 31
            and env("FNEUS_K3_R8", "1") != "0" and env("FNEUS_COL_BWD_R8", "1") != "0" and env("FNEUS_BWD_XHI", "1") != "0"
            and env("FNEUS_COLB_XHI", "1") != "0" and env("FNEUS_K3_HB") is None and env("FNEUS_BWD_WHI", "0") == "0")


DNORMAL_ACC = _os.environ.get("FNEUS_DNORMAL_ACC", "1") != "0"


def dnormal_accum_ok(n_pts: int, prec: int) -> bool:
    """may fneus_color_bwd ADD its gradient of the normals into a buffer that already holds the compositing backward's (autograd's sum
    of the two without a launch)?  Where the launch takes the resident-weight kernel (csrc/color_kernels.hip launch_bwd)"""
    tiles_pp = 2 * ((n_pts + 63) // 64)
    return (DNORMAL_ACC and prec in (PREC_PARITY, PREC_FAST) and (n_pts + 31) // 32 >= 1024 and tiles_pp * 4 * 16384 < 2 ** 31
            and _os.environ.get("FNEUS_COL_BWD_R8", "1") != "0")


def color_bwd(blob, n_pts, prec, d_rgb, rgb, stash: ColStash, head: int = HEAD_COLOR, normal=None, dirs=None, rays_d=None,
              m: int = 1, dfeat_plane=None, dn_accum=None):
    """dfeat_plane (head 0, only where dfeat_plane_ok): the bf16 fragment plane [tiles, 16, 64, 8] that receives the feature cotangent --
    the returned d_feat is then a PLACEHOLDER no kernel wrote (its `plane_of` attribute is the plane).
    dn_accum (head 0, only where dnormal_accum_ok): an fp32 [n, 3] tensor the gradient of the normals is ADDED into; the returned
    d_normal is then None"""
    _chk_f32(d_rgb, "d_rgb")
    d_feat = torch.empty(n_pts, 256, dtype=torch.float32, device=blob.device)
    if dn_accum is not None:
        if head != HEAD_COLOR or tuple(dn_accum.shape) != (n_pts, 3):
            raise ValueError("dn_accum: the colour network's backward only, an [n, 3] tensor")
        _chk_f32(dn_accum, "dn_accum")
    d_normal = torch.empty(n_pts, 3, dtype=torch.float32, device=blob.device) if dn_accum is None else dn_accum
    if head == HEAD_COLOR:
        stash.c.dnormal_add = 1 if dn_accum is not None else 0
        stash.c.dfeat_hi = dfeat_plane.data_ptr() if dfeat_plane is not None else None
        _launch("fneus_color_bwd", lib.fneus_color_bwd, _ptr(blob), n_pts, _ptr(d_rgb), _ptr(rgb), C.byref(stash.c),
                None if dfeat_plane is not None else _ptr(d_feat), _ptr(d_normal), prec, _stream())
        if dfeat_plane is not None:
            d_feat.plane_of = dfeat_plane
        if dn_accum is not None:
            return d_feat, None
    else:
        _chk_f32(normal, "normal")
        _launch("fneus_refcolor_bwd", lib.fneus_refcolor_bwd, _ptr(blob), head, n_pts, _ptr(rays_d), m, _ptr(dirs), _ptr(normal),
                _ptr(d_rgb), _ptr(rgb), C.byref(stash.c), _ptr(d_feat), _ptr(d_normal), prec, _stream())
    return d_feat, d_normal


def refcolor_fwd_both(blob_cd, blob_vd, n_pts, prec, normal, feat, stash_cd, stash_vd, train: bool, pts=None, rays_o=None,
                      rays_d=None, t=None, m: int = 1, dirs=None):
    """both RefColor heads in one launch -> diffuse [n,3], spec [n,3] (column 0)"""
    _chk_f32(normal, "normal")
    _chk_f32(feat, "feat")
    diffuse = torch.empty(n_pts, 3, dtype=torch.float32, device=feat.device)
    spec = torch.empty(n_pts, 3, dtype=torch.float32, device=feat.device)
    _launch("fneus_refcolor_fwd", lib.fneus_refcolor_fwd_both, _ptr(blob_cd), _ptr(blob_vd), _ptr(pts), _ptr(rays_o), _ptr(rays_d),
            _ptr(t), m, n_pts, _ptr(dirs), _ptr(normal), _ptr(feat), C.byref(stash_cd.c) if stash_cd is not None else None,
            C.byref(stash_vd.c) if stash_vd is not None else None, _ptr(diffuse), _ptr(spec), prec, int(train), _stream())
    return diffuse, spec


def refcolor_bwd_both(blob_cd, blob_vd, n_pts, prec, d_diffuse, d_spec, diffuse, spec, stash_cd, stash_vd, normal, dirs=None,
                      rays_d=None, m: int = 1):
    """-> d_feat2 [2,n,256], d_normal2 [2,n,3] (slice 0: diffuse head, slice 1: specular head)"""
    for x, nm in ((d_diffuse, "d_diffuse"), (d_spec, "d_spec"), (normal, "normal")):
        _chk_f32(x, nm)
    d_feat2 = torch.empty(2, n_pts, 256, dtype=torch.float32, device=normal.device)
    d_normal2 = torch.empty(2, n_pts, 3, dtype=torch.float32, device=normal.device)
    _launch("fneus_refcolor_bwd", lib.fneus_refcolor_bwd_both, _ptr(blob_cd), _ptr(blob_vd), n_pts, _ptr(rays_d), m, _ptr(dirs),
            _ptr(normal), _ptr(d_diffuse), _ptr(d_spec), _ptr(diffuse), _ptr(spec), C.byref(stash_cd.c), C.byref(stash_vd.c),
            _ptr(d_feat2), _ptr(d_normal2), prec, _stream())
    return d_feat2, d_normal2


class PPOperand:
    """One operand of a fragment-plane product: planes [P, tiles, F, 64, 8] bf16 (P = 1: hi, 2: hi + lo), the first
    fragment `f0` of the operand inside a block and its tile count; const = the same block for every sample tile."""

    def __init__(self, planes: torch.Tensor, f0: int, tiles: int, const: bool = False, lo: Optional[torch.Tensor] = None):
        """lo: the lo plane [tiles, F, 64, 8] as a tensor of its own (planes then holds the hi plane only)"""
        assert planes.dtype == torch.bfloat16 and planes.dim() == 5 and planes.shape[3:] == (64, 8), planes.shape
        assert planes.is_contiguous() or planes[0].is_contiguous()
        assert lo is None or (lo.is_contiguous() and lo.shape == planes.shape[1:] and lo.dtype == torch.bfloat16)
        self.planes, self.f0, self.tiles, self.const, self.lo = planes, f0, tiles, const, lo
        self.blk = 0 if const else planes.shape[2] * 1024
        assert f0 + 2 * tiles <= planes.shape[2], (f0, tiles, planes.shape)

    def ptr(self, p: int):
        if p == 1 and self.lo is not None:
            return self.lo.data_ptr()
        if p >= self.planes.shape[0]:
            return None
        assert self.planes[p].is_contiguous()
        return self.planes[p].data_ptr()


GEMM_COST_FLOOR = int(_os.environ.get("FNEUS_GEMM_FLOOR", "16"))     # in 32-row tiles of A + B (a full product: 16)


DET_TILE_FLOATS = 256 * 256 + 256          # csrc/dw_gemm_pp.hip kDetTile
_det_scratch = {}
_det_override = None


def deterministic() -> bool:
    """FNEUS_DETERMINISTIC=1 (or set_deterministic(True)): the weight-gradient GEMM sums its split-K partials in a fixed order
    instead of with fp32 atomics; every other kernel of the step is order-fixed already, so two runs are bit-identical.
    Read at every launch: set it before a step is captured into a hipGraph."""
    if _det_override is not None:
        return _det_override
    return os.environ.get("FNEUS_DETERMINISTIC", "0") not in ("", "0")


def set_deterministic(on: Optional[bool]):
    global _det_override
    _det_override = on


class GemmPPJobs:
    """Device job table for fneus_dw_gemm_pp (include/fneus.h FneusGemmPPJob); pointers refer to live plane tensors."""

    def __init__(self, device, tag="", target_wgs: int = 256, cost_floor: Optional[int] = None):
        self.device, self.tag, self.target = device, tag, target_wgs
        self.cost_floor = GEMM_COST_FLOOR if cost_floor is None else cost_floor
        self.jobs, self.bytes, self.live = [], [], []
        # share of their tiles that products with a device-side sample count are expected to sum (the host does not know the
        # count when it hands out the workgroups): the background network's list holds its n_out samples per ray and the few of
        # the other n that lie outside the unit sphere -- a quarter to a third of n + n_out
        self.live_fraction = 0.3
        self.dev_table, self.n_wgs = None, 0

    def add(self, A: PPOperand, B: PPOperand, c_ptr, ldc, m, n, A2: Optional[PPOperand] = None, B2: Optional[PPOperand] = None,
            bias_ptr=None, scale=1.0, n_tiles: int = 0, n_dev: Optional[torch.Tensor] = None):
        """n_tiles: sample tiles of this product's planes when they differ from the launch's (finalize); n_dev: device int32 count
        of the samples they hold this step (the tensor must outlive the table)"""
        assert A.tiles <= 8 and B.tiles <= 8 and m <= 32 * A.tiles and n <= 32 * B.tiles
        j = _lib.FneusGemmPPJob()
        j.a_hi, j.a_lo, j.b_hi, j.b_lo = A.ptr(0), A.ptr(1), B.ptr(0), B.ptr(1)
        j.a_blk, j.b_blk, j.a_f0, j.b_f0 = A.blk, B.blk, A.f0, B.f0
        if A2 is not None:
            assert A2.tiles == A.tiles and B2.tiles == B.tiles
            j.a2_hi, j.a2_lo, j.b2_hi, j.b2_lo = A2.ptr(0), A2.ptr(1), B2.ptr(0), B2.ptr(1)
            j.a2_blk, j.b2_blk, j.a2_f0, j.b2_f0 = A2.blk, B2.blk, A2.f0, B2.f0
        j.mt, j.nt, j.c, j.bias, j.ldc, j.m, j.n, j.scale = A.tiles, B.tiles, c_ptr, bias_ptr, ldc, m, n, scale
        j.n_tiles = int(n_tiles)
        j.n_dev = None if n_dev is None else n_dev.data_ptr()
        self.live.append(1.0 if n_dev is None else self.live_fraction)
        self.jobs.append(j)
        # cost of one sample tile of this job, for the workgroup distribution.  The kernel's work per stage does not shrink
        # with a narrow operand (a stage always DMAs and multiplies full 16-fragment parts; there are no branches in its
        # loop), so every product costs the same: floor = 16 tiles.  Measured (A/B on one box, floors 0 / 8 / 12 / 16): SDF
        # 0.31 / 0.30 / 0.25 / 0.25 ms, background NeRF (jobs as narrow as 1 + 4 tiles) 0.43 / 0.27 / 0.22 / 0.21 ms.
        self.bytes.append(max(A.tiles + B.tiles, self.cost_floor) * 2048 * (2 if A2 is not None else 1))
        return self

    def finalize(self, n_sample_tiles: int, min_tiles: int = 4):
        """distribute ~target workgroups over the jobs in proportion to the bytes each streams; a workgroup owns at least
        `min_tiles` sample tiles (its epilogue is up to 65 536 atomics whatever it summed)"""
        tiles = [j.n_tiles if j.n_tiles > 0 else n_sample_tiles for j in self.jobs]
        cost = [b * t * f for b, t, f in zip(self.bytes, tiles, self.live)]
        tot = float(sum(cost))
        splits = [max(1, min(max(1, t // min_tiles), int(round(self.target * c / tot)))) for c, t in zip(cost, tiles)]
        # the kernel keeps ONE workgroup per CU (128 KB of LDS): a launch of more than `target` workgroups needs a second
        # round for the few extra ones and takes nearly twice as long (measured: 258 workgroups 0.39 ms, 238 0.29 ms).
        # Rounding must therefore never push the total over the target: trim the jobs with the most splits.
        while sum(splits) > self.target and max(splits) > 1:
            splits[max(range(len(splits)), key=lambda i: splits[i])] -= 1
        base = 0
        for j, s in zip(self.jobs, splits):
            j.wg_base, j.splits = base, s
            base += s
        self.n_wgs, self.n_sample_tiles = base, n_sample_tiles
        arr = (_lib.FneusGemmPPJob * len(self.jobs))(*self.jobs)
        self.dev_table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
        return self

    def run(self, *_ignored, gprec: Optional[int] = None, n_dev: Optional[torch.Tensor] = None):
        """(positional arguments are accepted and ignored: older callers passed n and prec)
        n_dev: device int32 sample count (outside_select): only the sample tiles of the first n_dev samples are summed"""
        gprec = getattr(self, "gprec", 1) if gprec is None else gprec
        if deterministic():
            # bit-reproducible gradients: partial tiles to a scratch buffer, summed in split order by a second launch (the
            # buffer is per device and shared by all job tables: launches on a stream do not overlap)
            need = self.n_wgs * DET_TILE_FLOATS
            buf = _det_scratch.get(self.device)
            if buf is None or buf.numel() < need:
                buf = _det_scratch[self.device] = torch.empty(max(need, 256 * DET_TILE_FLOATS), dtype=torch.float32, device=self.device)
            _launch("fneus_dw_gemm_pp_det:" + self.tag, lib.fneus_dw_gemm_pp_det, _ptr(self.dev_table), len(self.jobs), self.n_wgs,
                    self.n_sample_tiles, _ptr(n_dev), gprec, _ptr(buf), buf.numel(), _stream())
            return
        _launch("fneus_dw_gemm_pp:" + self.tag, lib.fneus_dw_gemm_pp, _ptr(self.dev_table), len(self.jobs), self.n_wgs,
                self.n_sample_tiles, _ptr(n_dev), gprec, _stream())


class _OwnPlanes:
    """GemmPPJobs.add for products over another network's planes in a shared launch: their own sample-tile count and, where the
    planes hold a different number of samples every step, their device-side count"""

    def __init__(self, jobs: GemmPPJobs, n_tiles: int, n_dev: Optional[torch.Tensor] = None):
        self.jobs, self.n_tiles, self.n_dev = jobs, n_tiles, n_dev

    def add(self, *a, **k):
        return self.jobs.add(*a, n_tiles=self.n_tiles, n_dev=self.n_dev, **k)


def gemm_merge_enabled() -> bool:
    """FNEUS_GEMM_MERGE=0: the colour network's weight-gradient products as a launch of their own behind its backward (comparison
    runs); default: in the SDF network's launch behind K3 (one launch over the same 65 536 samples instead of two)"""
    return os.environ.get("FNEUS_GEMM_MERGE", "1") != "0"


def sdf_dw_jobs(net: PackedNet, stash: SdfStash, bufs: SdfBwdBufs, grad_flat: torch.Tensor, n: int, also=None) -> GemmPPJobs:
    """dW_l = zbar_l^T u_l + a_l^T adj_l for the 9 SDF layers, into the flat fp32 grad buffer;
    operands are the fragment planes written by K2 / K3.  also(table): products over the same samples to append (the colour
    network's, color_dw_jobs(..., into=table)) before the workgroups are distributed."""
    import math
    g = GemmPPJobs(grad_flat.device, "sdf" if also is None else "sdf+color")       # ("+color": and whatever else waited)
    O = PPOperand
    offW, offb = net.desc["offW"], net.desc["offb"]
    base = grad_flat.data_ptr()
    isq2 = 1.0 / math.sqrt(2.0)
    zb, h, a, adj = (lambda l: bufs.zbar[:, l]), (lambda l: stash.h[:, l]), (lambda l: stash.a[:, l]), (lambda l: bufs.adj[:, l])
    for l in (1, 2, 3, 5, 6, 7):
        mt = 7 if l == 3 else 8
        g.add(O(zb(l), 0, mt), O(h(l - 1), 0, 8), base + 4 * offW[l], 256, 217 if l == 3 else 256, 256,
              A2=O(a(l), 0, mt), B2=O(adj(l - 1), 0, 8), bias_ptr=base + 4 * offb[l])
    # layer 0: inputs = PE
    g.add(O(zb(0), 0, 8), O(stash.pe, 0, 2), base + 4 * offW[0], 39, 256, 39,
          A2=O(a(0), 0, 8), B2=O(bufs.qbar, 0, 2), bias_ptr=base + 4 * offb[0])
    # layer 4: [h_4 (217) ; PE (39)] / sqrt(2)
    g.add(O(zb(4), 0, 8), O(h(3), 0, 7), base + 4 * offW[4], 256, 256, 217,
          A2=O(a(4), 0, 8), B2=O(adj(3), 0, 7), bias_ptr=base + 4 * offb[4], scale=isq2)
    g.add(O(zb(4), 0, 8), O(stash.pe, 0, 2), base + 4 * (offW[4] + 217), 256, 256, 39,
          A2=O(a(4), 0, 8), B2=O(bufs.qbar, 0, 2), scale=isq2)
    # layer 8: rows 1..256 (feature), row 0 (sdf; its ascending term is sum_n adj_8[n]: a_8 = e_0, a constant block)
    g.add(O(zb(8), 0, 8), O(h(7), 0, 8), base + 4 * (offW[8] + 256), 256, 256, 256, bias_ptr=base + 4 * (offb[8] + 1))
    g.add(O(bufs.zsdf, 0, 1), O(h(7), 0, 8), base + 4 * offW[8], 256, 1, 256,
          A2=O(bufs.ones, 0, 1, const=True), B2=O(adj(7), 0, 8), bias_ptr=base + 4 * offb[8])
    g.gprec = stash.gprec
    if also is not None:
        also(g)
        assert g.gprec == stash.gprec, "merged weight-gradient products differ in gradient precision"
    return g.finalize(stash.tiles)


def color_dw_jobs(net: PackedNet, feat_planes: torch.Tensor, stash: ColStash, grad_flat: torch.Tensor, n: int,
                  into: Optional[GemmPPJobs] = None, own_tiles: bool = False) -> GemmPPJobs:
    """dW of a colour-shaped MLP (colour network: feat_planes = SdfStash.feat; RefColor heads: their own ColStash.feat)
    from fragment planes.  `into`: append to that job table instead of finalising a new one (several small networks in
    one launch)."""
    g = into if into is not None else GemmPPJobs(grad_flat.device, net.kind)
    O = PPOperand
    offW, offb = net.desc["offW"], net.desc["offb"]
    n_side, ld0, n_out = net.desc["n_side"], net.desc["ins"][0], net.desc["outs"][4]
    base = grad_flat.data_ptr()
    # (the SDF stash keeps the feature planes hi + lo in parity mode -- they are the colour network's input; the products take as many
    #  planes as the colour stash's gradient precision has)
    assert feat_planes.shape[0] >= stash.zbar.shape[0], "feature planes and colour stash differ in gradient precision"
    feat_planes = feat_planes[:stash.zbar.shape[0]]
    nt = stash.tiles if own_tiles else 0         # own_tiles: these planes hold another number of samples than the launch's
    # layer 0: columns 0..n_side-1 = side inputs, then the 256 features
    g.add(O(stash.zbar[:, 0], 0, 8), O(feat_planes, 0, 8), base + 4 * (offW[0] + n_side), ld0, 256, 256, bias_ptr=base + 4 * offb[0],
          n_tiles=nt)
    g.add(O(stash.zbar[:, 0], 0, 8), O(stash.side, 0, 2), base + 4 * offW[0], ld0, 256, n_side, n_tiles=nt)
    for l in (1, 2, 3):
        g.add(O(stash.zbar[:, l], 0, 8), O(stash.u[:, l - 1], 0, 8), base + 4 * offW[l], 256, 256, 256, bias_ptr=base + 4 * offb[l],
              n_tiles=nt)
    if stash.gprec != 2:      # (gradient precision 2: the output layer's product runs on hi + lo planes in a launch of its own,
        g.add(O(stash.zout, 0, 1), O(stash.u[:, 3], 0, 8), base + 4 * offW[4], 256, n_out, 256, bias_ptr=base + 4 * offb[4], n_tiles=nt)
    g.gprec = 1 if stash.gprec == 2 else stash.gprec                                             # color_out_dw_jobs)
    return g if into is not None else g.finalize(stash.tiles)


def color_out_dw_jobs(net: PackedNet, stash: ColStash, grad_flat: torch.Tensor) -> GemmPPJobs:
    """gradient precision 2: dW and db of the colour network's output layer (3 x 256) from the hi + lo planes of its two operands --
    a launch of its own at gradient precision 3 (a launch is one precision); 0.6 KB per sample against the 2 KB of all lo planes"""
    assert stash.gprec == 2
    g = GemmPPJobs(grad_flat.device, net.kind + ".out")
    offW, offb, n_out = net.desc["offW"], net.desc["offb"], net.desc["outs"][4]
    base = grad_flat.data_ptr()
    g.add(PPOperand(stash.zout, 0, 1, lo=stash.zout_lo), PPOperand(stash.u[:, 3], 0, 8, lo=stash.u3_lo), base + 4 * offW[4], 256, n_out, 256,
          bias_ptr=base + 4 * offb[4])
    g.gprec = 3
    return g.finalize(stash.tiles)


def color_out_dw(net: PackedNet, stash: ColStash, d_rgb: torch.Tensor, rgb: torch.Tensor, grad_flat: torch.Tensor, n: int, cache=None):
    """gradient precision 2: dW and db of the colour network's output layer from the hi + lo planes of u_3 and zout formed in fp32 from
    d_rgb and rgb (fneus_color_out_dw: one streaming pass, 1 KiB per sample).  The deterministic mode keeps the fixed-order GEMM on
    the hi + lo planes of both operands (its job table lives in `cache`, a workspace: a captured step keeps its address)."""
    assert stash.gprec == 2
    if deterministic():
        make = lambda: color_out_dw_jobs(net, stash, grad_flat)
        jobs = make() if cache is None else cache.get(("col_out_jobs", n, stash.zout.data_ptr(), grad_flat.data_ptr()), make)
        return jobs.run()
    offW, offb = net.desc["offW"], net.desc["offb"]
    _chk_f32(d_rgb, "d_rgb")
    _chk_f32(rgb, "rgb")
    if getattr(stash, "out_dw_scratch", None) is None:      # the replicas of the 771 sums: zero once, every call leaves them zero
        stash.out_dw_scratch = torch.zeros(int(lib.fneus_color_out_dw_scratch_floats()), dtype=torch.float32, device=grad_flat.device)
    _launch("fneus_color_out_dw", lib.fneus_color_out_dw, _ptr(stash.u[0, 3]), _ptr(stash.u3_lo), _ptr(d_rgb), _ptr(rgb), int(n),
            C.c_void_p(grad_flat.data_ptr() + 4 * offW[4]), C.c_void_p(grad_flat.data_ptr() + 4 * offb[4]), _ptr(stash.out_dw_scratch),
            *_take_fold_rider(grad_flat.device), _stream())


# A small reduction that waits for the next one-workgroup fold launch of the backward pass (fneus_color_out_dw's): CompositeFn.backward
# leaves the per-ray gradients of inv_s here with the address of the variance parameter's gradient; whatever still waits when the
# backward pass ends is summed by torch (autograd-engine callback).  One slot per device.
_FOLD_RIDER = {}
DEFAULT_FOLD_RIDER = _os.environ.get("FNEUS_FOLD_RIDER", "1") != "0"


def offer_fold_rider(src: torch.Tensor, dst: torch.Tensor):
    """sum(src) is to be ADDED to the one-element tensor dst by the next fold launch on this device"""
    dev = src.device
    rec = (src, dst)
    _FOLD_RIDER[dev] = rec

    def flush():
        if _FOLD_RIDER.get(dev) is rec:
            del _FOLD_RIDER[dev]
            dst.add_(src.sum())

    torch.autograd.Variable._execution_engine.queue_callback(flush)


def flush_fold_rider(dev):
    """apply a rider nobody has taken (SdfValueGradFn.backward calls this in front of the data-parallel step's early gradient
    exchange: the variance gradient has to be complete by then)"""
    rec = _FOLD_RIDER.pop(dev, None)
    if rec is not None:
        rec[1].add_(rec[0].sum())


def _take_fold_rider(dev):
    rec = _FOLD_RIDER.pop(dev, None)
    if rec is None:
        return None, 0, None
    src, dst = rec
    return _ptr(src), int(src.numel()), _ptr(dst)


# ------------------------------------------------------------------------------------------------------------
# K7: background NeRF++ (womask)
# ------------------------------------------------------------------------------------------------------------
class NerfStash:
    """fragment planes of fneus_nerf_bg_fwd / _bwd (include/fneus.h FneusNerfStash): [P, (layer,) tiles, F, 64, 8] bf16,
    P = 1 (hi; gradient precision 1) or 2 (hi + lo)"""

    def __init__(self, n: int, device, prec: int, gprec: Optional[int] = None):
        from . import pp
        bf = torch.bfloat16
        self.gprec = _gprec(prec, gprec)
        planes = 2 if self.gprec == 3 else 1
        T = pp.alloc_tiles(n)
        self.tiles = pp.n_tiles(n)
        z = lambda *shape: torch.zeros((planes,) + shape + (64, 8), dtype=bf, device=device)
        self.pe, self.h, self.feat, self.dpe, self.hv = z(T, 6), z(8, T, 16), z(T, 16), z(T, 2), z(T, 8)
        self.zbar, self.zfeat, self.zhv, self.zout = z(8, T, 16), z(T, 16), z(T, 8), z(T, 4)
        self.mask = torch.zeros(((n + 31) // 32) * 9 * 64 * 4, dtype=torch.int32, device=device)
        s = _lib.FneusNerfStash()
        s.mask = self.mask.data_ptr()
        for name in ("pe", "h", "feat", "dpe", "hv", "zbar", "zfeat", "zhv", "zout"):
            t = getattr(self, name)
            setattr(s, name + "_hi", t[0].data_ptr())
            setattr(s, name + "_lo", t[1].data_ptr() if planes == 2 else None)
        self.c = s


def nerf_fwd(blob, n_pts, prec, pts4, dirs, stash: Optional[NerfStash], train: bool, n_dev: Optional[torch.Tensor] = None):
    """NeRF.forward (fields.py:233-259) on the HIP engine -> raw density [n], raw rgb [n,3]
    n_dev: device int32 count (outside_select): only the first n_dev rows are evaluated, the others are left unwritten"""
    _chk_f32(pts4, "pts4")
    _chk_f32(dirs, "dirs")
    density = torch.empty(n_pts, dtype=torch.float32, device=blob.device)
    rgb = torch.empty(n_pts, 3, dtype=torch.float32, device=blob.device)
    _launch("fneus_nerf_bg_fwd", lib.fneus_nerf_bg_fwd, _ptr(blob), _ptr(pts4), _ptr(dirs), n_pts,
            C.byref(stash.c) if stash is not None else None, _ptr(density), _ptr(rgb), prec, int(train), _ptr(n_dev), _stream())
    return density, rgb


def nerf_bwd(blob, n_pts, prec, d_density, d_rgb, stash: NerfStash, n_dev: Optional[torch.Tensor] = None):
    _chk_f32(d_density, "d_density")
    _chk_f32(d_rgb, "d_rgb")
    _launch("fneus_nerf_bg_bwd", lib.fneus_nerf_bg_bwd, _ptr(blob), n_pts, _ptr(d_density), _ptr(d_rgb), C.byref(stash.c), prec,
            _ptr(n_dev), _stream())


def nerf_dw_jobs(net: PackedNet, st: NerfStash, n: int, into: Optional[GemmPPJobs] = None,
                 n_dev: Optional[torch.Tensor] = None) -> GemmPPJobs:
    """weight / bias gradients of the 12 Linear layers from the fragment planes, accumulated straight into net.raw_grad
    (plain layers: the raw layout IS the effective one).  into: append to that table (products with their own tile count and,
    n_dev given, their own device-side sample count) instead of finalising a new one"""
    g = _OwnPlanes(into, st.tiles, n_dev) if into is not None else GemmPPJobs(net.raw_grad.device, "nerf")
    O = PPOperand
    d = net.desc
    P = {name: i for i, name in enumerate(d["names"])}
    base = net.raw_grad.data_ptr()
    W = lambda name, col=0: base + 4 * (d["offV"][P[name]] + col)
    Bv = lambda name: base + 4 * d["offB"][P[name]]
    zb = lambda l: st.zbar[:, l]
    hh = lambda l: st.h[:, l]
    g.add(O(zb(0), 0, 8), O(st.pe, 0, 3), W("pts_linears.0"), 84, 256, 84, bias_ptr=Bv("pts_linears.0"))
    for l in (1, 2, 3, 4, 6, 7):
        g.add(O(zb(l), 0, 8), O(hh(l - 1), 0, 8), W(f"pts_linears.{l}"), 256, 256, 256, bias_ptr=Bv(f"pts_linears.{l}"))
    # pts_linears.5: columns [PE 84 | h 256] (fields.py:245)
    g.add(O(zb(5), 0, 8), O(st.pe, 0, 3), W("pts_linears.5"), 340, 256, 84, bias_ptr=Bv("pts_linears.5"))
    g.add(O(zb(5), 0, 8), O(hh(4), 0, 8), W("pts_linears.5", 84), 340, 256, 256)
    g.add(O(st.zfeat, 0, 8), O(hh(7), 0, 8), W("feature_linear"), 256, 256, 256, bias_ptr=Bv("feature_linear"))
    g.add(O(st.zout, 2, 1), O(hh(7), 0, 8), W("alpha_linear"), 256, 1, 256, bias_ptr=Bv("alpha_linear"))
    # views_linears.0: columns [feature 256 | PE4(view) 27]
    g.add(O(st.zhv, 0, 4), O(st.feat, 0, 8), W("views_linears.0"), 283, 128, 256, bias_ptr=Bv("views_linears.0"))
    g.add(O(st.zhv, 0, 4), O(st.dpe, 0, 1), W("views_linears.0", 256), 283, 128, 27)
    g.add(O(st.zout, 0, 1), O(st.hv, 0, 4), W("rgb_linear"), 128, 3, 128, bias_ptr=Bv("rgb_linear"))
    if into is not None:
        return into
    g.gprec = st.gprec
    return g.finalize(st.tiles)


# ------------------------------------------------------------------------------------------------------------
# per-ray kernels
# ------------------------------------------------------------------------------------------------------------
def surface_gather(min_idx, sdf_mask, mid_z, feat, normal, warm=None):
    """-> sel [2B] int32, t_sel [2B], feat_sel [2B,256], normal_sel [2B,3];  warm: WarmRanges of the launch that follows"""
    B, n = mid_z.shape
    dev = mid_z.device
    sel = torch.empty(2 * B, dtype=torch.int32, device=dev)
    t_sel = torch.empty(2 * B, dtype=torch.float32, device=dev)
    feat_sel = torch.empty(2 * B, 256, dtype=torch.float32, device=dev)
    normal_sel = torch.empty(2 * B, 3, dtype=torch.float32, device=dev)
    src = getattr(feat, "planes_of", None)          # a placeholder of sdf_fwd_grad(feat_rows=False): the rows are read from the planes
    _launch("fneus_surface_gather", lib.fneus_surface_gather, _ptr(min_idx), _ptr(sdf_mask), _ptr(mid_z),
            None if src is not None else _ptr(feat), None if src is None else _ptr(src.feat[0]),
            None if (src is None or src.feat.shape[0] < 2) else _ptr(src.feat[1]), _ptr(normal),
            B, n, _ptr(sel), _ptr(t_sel), _ptr(feat_sel), _ptr(normal_sel), _warm_arg(warm), _stream())
    return sel, t_sel, feat_sel, normal_sel


def surface_scatter(sel, d_feat_heads, d_normal_heads, d_feat, d_normal):
    """d_feat[sel] += sum over heads of d_feat_heads [H, R, 256]; d_normal[sel] += sum of d_normal_heads [H, R, 3] (in place)"""
    H, R = (d_feat_heads if d_feat_heads is not None else d_normal_heads).shape[:2]
    for x, nm in ((d_feat_heads, "d_feat_heads"), (d_normal_heads, "d_normal_heads"), (d_feat, "d_feat"), (d_normal, "d_normal")):
        if x is not None:
            _chk_f32(x, nm)
    _launch("fneus_surface_scatter", lib.fneus_surface_scatter, _ptr(sel), _ptr(d_feat_heads), _ptr(d_normal_heads), int(H), int(R),
            _ptr(d_feat), _ptr(d_normal), _stream())


def surface_scatter_plane(sel, d_feat_heads, d_normal_heads, dfeat_plane, n_pts, d_normal):
    """surface_scatter with the feature cotangent as bf16 fragments [tiles, 16, 64, 8] (csrc/fneus_pp.h) instead of rows"""
    H, R = (d_feat_heads if d_feat_heads is not None else d_normal_heads).shape[:2]
    for x, nm in ((d_feat_heads, "d_feat_heads"), (d_normal_heads, "d_normal_heads"), (d_normal, "d_normal")):
        if x is not None:
            _chk_f32(x, nm)
    if dfeat_plane.dtype != torch.bfloat16 or not dfeat_plane.is_contiguous():
        raise ValueError("dfeat_plane: contiguous bf16 fragments expected")
    _launch("fneus_surface_scatter_plane", lib.fneus_surface_scatter_plane, _ptr(sel), _ptr(d_feat_heads), _ptr(d_normal_heads), int(H), int(R),
            _ptr(dfeat_plane), int(n_pts), _ptr(d_normal), _stream())


def stage1_norms(mask_in, sdf_mask, eik_den, mask_weight):
    """[sum mask, sum mask*sdf_mask, sum eik_den, ray count] of this batch (to be summed over the ranks)"""
    norms = torch.empty(4, dtype=torch.float32, device=mask_in.device)
    _launch("fneus_stage1_norms", lib.fneus_stage1_norms, _ptr(mask_in), _ptr(sdf_mask), _ptr(eik_den), int(mask_in.numel()),
            float(mask_weight), _ptr(norms), _stream())
    return norms


def stage1_loss(color, true_rgb, mask_in, wsum, eik_num, eik_den, diffuse, spec, wpair, sdf_mask, igr_weight, mask_weight,
                surface_weight, norms=None, warm=None):
    """fused shading + losses + gradients (include/fneus.h fneus_stage1_loss); returns a dict of device tensors"""
    B = color.shape[0]
    dev = color.device
    f32 = dict(dtype=torch.float32, device=dev)
    for x, nm in ((color, "color"), (true_rgb, "true_rgb"), (mask_in, "mask"), (wsum, "wsum"), (eik_num, "eik_num"),
                  (eik_den, "eik_den"), (diffuse, "diffuse"), (spec, "spec"), (wpair, "wpair")):
        _chk_f32(x, nm)
    both = torch.empty(9, **f32)          # losses[8] = the total once more: the loss tensor of its own (include/fneus.h)
    o = {"losses": both[:8], "loss": both[8:].view(()), "surface_color": torch.empty(B, 3, **f32), "specular_color": torch.empty(B, 3, **f32),
         "diffuse_color": torch.empty(B, 3, **f32), "d_color": torch.empty(B, 3, **f32), "d_wsum": torch.empty(B, **f32),
         "d_eiknum": torch.empty(B, **f32), "d_wpair": torch.empty(B, 2, **f32), "d_diffuse": torch.empty(2 * B, 3, **f32),
         "d_spec": torch.empty(2 * B, 3, **f32)}
    _launch("fneus_stage1_loss", lib.fneus_stage1_loss, _ptr(color), _ptr(true_rgb), _ptr(mask_in), _ptr(wsum), _ptr(eik_num),
            _ptr(eik_den), _ptr(diffuse), _ptr(spec), _ptr(wpair), _ptr(sdf_mask), _ptr(norms), B, float(igr_weight), float(mask_weight),
            float(surface_weight), _ptr(o["losses"]), _ptr(o["surface_color"]), _ptr(o["specular_color"]),
            _ptr(o["diffuse_color"]), _ptr(o["d_color"]), _ptr(o["d_wsum"]), _ptr(o["d_eiknum"]), _ptr(o["d_wpair"]),
            _ptr(o["d_diffuse"]), _ptr(o["d_spec"]), _warm_arg(warm), _stream())
    return o


def upsample(rays_o, rays_d, z, sdf, k: int, inv_s: float):
    B, m = z.shape
    out = torch.empty(B, k, dtype=torch.float32, device=z.device)
    _launch("fneus_upsample", lib.fneus_upsample, _ptr(rays_o), _ptr(rays_d), _ptr(z), _ptr(sdf), B, m, k, float(inv_s), _ptr(out),
                             _stream())
    return out


class _SrgbFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = x.contiguous()
        y = torch.empty_like(x)
        _launch("fneus_srgb_fwd", lib.fneus_srgb_fwd, _ptr(x), x.numel(), mode, _ptr(y), _stream())
        ctx.save_for_backward(x)
        ctx.mode = mode
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _launch("fneus_srgb_bwd", lib.fneus_srgb_bwd, _ptr(x), _ptr(dy), x.numel(), ctx.mode, _ptr(dx), _stream())
        return dx, None


def srgb(x, to_linear: bool = False, clip: bool = False):
    """linear_to_srgb / srgb_to_linear (math_utils.py:138-152), optionally followed by clip(., 0, 1), as one launch (fneus_srgb_fwd)"""
    return _SrgbFn.apply(x, (1 if to_linear else 0) | (2 if clip else 0))


def vis_sample_dirs(lobes, lambdas, u_theta, u_phi):
    """fneus_vis_sample_dirs: lobes [M,3], lambdas [M], u_theta / u_phi [M,S] -> dirs [M,S,3], weights [M,S]"""
    M, S = u_theta.shape
    dirs = torch.empty(M, S, 3, dtype=torch.float32, device=lobes.device)
    w = torch.empty(M, S, dtype=torch.float32, device=lobes.device)
    _launch("fneus_vis_sample_dirs", lib.fneus_vis_sample_dirs, _ptr(lobes), _ptr(lambdas), _ptr(u_theta), _ptr(u_phi), M, S, _ptr(dirs),
            _ptr(w), _stream())
    return dirs, w


def vis_sample_dirs_sgs(lgt_sgs, u_theta, u_phi):
    """fneus_vis_sample_dirs_sgs: the light-SG table lgtSGs [M,7] itself (axis normalisation and |sharpness| inside the launch),
    u_theta / u_phi [M,S] -> dirs [M,S,3], weights [M,S]"""
    M, S = u_theta.shape
    dirs = torch.empty(M, S, 3, dtype=torch.float32, device=lgt_sgs.device)
    w = torch.empty(M, S, dtype=torch.float32, device=lgt_sgs.device)
    _launch("fneus_vis_sample_dirs_sgs", lib.fneus_vis_sample_dirs_sgs, _ptr(lgt_sgs), _ptr(u_theta), _ptr(u_phi), M, S, _ptr(dirs), _ptr(w),
            _stream())
    return dirs, w


def material_inputs(points, ray_dirs, normals):
    """fneus_material_inputs (inverRender.py:530-545, no gradient): points, ray_dirs, normals [n,3] -> unit normal [n,3], view
    direction [n,3], embed(points, 10) [n,63], [embed(points, 10) | embed(reflected direction, 4)] [n,90]"""
    n = points.shape[0]
    for t, name in ((points, "points"), (ray_dirs, "ray_dirs"), (normals, "normals")):
        _chk_f32(t, name)
    mk = lambda w: torch.empty(n, w, dtype=torch.float32, device=points.device)
    n_unit, view, enc, x_cs = mk(3), mk(3), mk(63), mk(90)
    _launch("fneus_material_inputs", lib.fneus_material_inputs, _ptr(points), _ptr(ray_dirs), _ptr(normals), n, _ptr(n_unit), _ptr(view),
            _ptr(enc), _ptr(x_cs), _stream())
    return n_unit, view, enc, x_cs


def indir_sgs(raw):
    """fneus_indir_sgs: IndirectLight's output transform without gradient, raw [n, L, 6] -> [n, L, 7]"""
    raw = raw.contiguous()
    out = torch.empty(raw.shape[0], raw.shape[1], 7, dtype=torch.float32, device=raw.device)
    _launch("fneus_indir_sgs", lib.fneus_indir_sgs, _ptr(raw), raw.shape[0] * raw.shape[1], _ptr(out), _stream())
    return out


def indir_illum_fwd(raw, dirs):
    """raw [n, L, 6] (IndirectLight's MLP output), dirs [n, S, 3] -> radiance [n, S, 3] (fneus_indir_illum_fwd)"""
    n, L, S = raw.shape[0], raw.shape[1], dirs.shape[1]
    out = torch.empty(n, S, 3, dtype=torch.float32, device=raw.device)
    _launch("fneus_indir_illum_fwd", lib.fneus_indir_illum_fwd, _ptr(raw), _ptr(dirs), n, L, S, _ptr(out), _stream())
    return out


def indir_illum_bwd(raw, dirs, d_rad):
    n, L, S = raw.shape[0], raw.shape[1], dirs.shape[1]
    d_raw = torch.empty_like(raw)
    _launch("fneus_indir_illum_bwd", lib.fneus_indir_illum_bwd, _ptr(raw), _ptr(dirs), _ptr(d_rad), n, L, S, _ptr(d_raw), _stream())
    return d_raw


# ---- the trained plain MLPs of stages 2 / 3 on a few hundred rows (fneus_mlp_*: csrc/mlp_rows_kernels.hip) -------------------------
ACT_NONE, ACT_RELU, ACT_LEAKY02, ACT_SIGMOID = 0, 1, 2, 3
MLP_MAX_JOBS = 16
# FNEUS_MLP_ROWS=0: the stage-2 / 3 modules keep torch's nn.Sequential (rocBLAS GEMMs + element-wise launches), for A/B runs
MLP_ROWS = os.environ.get("FNEUS_MLP_ROWS", "1") != "0"


def _mlp_jobs(jobs):
    from ._lib import FneusMlpJob
    arr = (FneusMlpJob * len(jobs))()
    for a, j in zip(arr, jobs):
        for k in ("x", "weight", "bias", "y", "dy", "dx", "d_weight", "d_bias"):
            t = j.get(k)
            if t is not None:
                _chk_f32(t, f"fneus_mlp: {k}")
                setattr(a, k, t.data_ptr())
        a.rows, a.n_in, a.n_out = int(j["rows"]), int(j["n_in"]), int(j["n_out"])
        a.act, a.act_in = int(j.get("act", 0)), int(j.get("act_in", 0))
    return arr


def mlp_forward(jobs):
    """one launch: y = act(x W^T + b) for every job (dicts with the fields of FneusMlpJob, include/fneus.h)"""
    _launch("fneus_mlp_forward", lib.fneus_mlp_forward, _mlp_jobs(jobs), len(jobs), _stream())


def mlp_backward_input(jobs):
    """one launch: dx = ((dy * act'(y)) W) * act_in'(x) for every job"""
    _launch("fneus_mlp_backward_input", lib.fneus_mlp_backward_input, _mlp_jobs(jobs), len(jobs), _stream())


def mlp_backward_params(jobs):
    """one launch: d_weight = (dy * act'(y))^T x and d_bias = its column sums for every job (both overwritten)"""
    _launch("fneus_mlp_backward_params", lib.fneus_mlp_backward_params, _mlp_jobs(jobs), len(jobs), _stream())


def outside_z(rays_o, rays_d, n_outside: int, n_samples: int, far=None, u=None):
    """z_vals_outside of NeuSRenderer.render (renderer.py:397-400, 411-419) -> [B, n_outside]; far [B] or None (unit-sphere bound
    of the rays), u [B, n_outside] uniform draws or None (no jitter)"""
    B = rays_o.shape[0]
    z = torch.empty(B, n_outside, dtype=torch.float32, device=rays_o.device)
    _launch("fneus_outside_z", lib.fneus_outside_z, _ptr(rays_o), _ptr(rays_d), _ptr(far), _ptr(u), B, n_outside, n_samples, _ptr(z),
            _stream())
    return z


def outside_points(rays_o, rays_d, z_feed, sample_dist: float):
    """render_core_outside's geometry (renderer.py:121-131) -> pts4 [B*nt,4], dirs [B*nt,3], dists [B,nt]"""
    B, nt = z_feed.shape
    dev = z_feed.device
    pts4 = torch.empty(B * nt, 4, dtype=torch.float32, device=dev)
    dirs = torch.empty(B * nt, 3, dtype=torch.float32, device=dev)
    dists = torch.empty(B, nt, dtype=torch.float32, device=dev)
    _launch("fneus_outside_points", lib.fneus_outside_points, _ptr(rays_o), _ptr(rays_d), _ptr(z_feed), B, nt, float(sample_dist),
            _ptr(pts4), _ptr(dirs), _ptr(dists), _stream())
    return pts4, dirs, dists


class OutsideSelection:
    """what fneus_outside_select lists (include/fneus.h): the background samples whose value render_core uses"""
    __slots__ = ("pts4", "dirs", "dists", "sel", "count", "alpha_full", "rgb_full", "B", "nt", "cap")


def bg_select_enabled() -> bool:
    """FNEUS_BG_SELECT=0: the background network at every merged depth, as the reference evaluates it (comparison runs)"""
    return os.environ.get("FNEUS_BG_SELECT", "1") != "0"


def outside_select(rays_o, rays_d, z_core, z_feed, sample_dist: float, count: Optional[torch.Tensor] = None) -> OutsideSelection:
    """count: int32[1] buffer for the list's length (a persistent one where tables hold its address: the weight-gradient job table
    of a training step); a fresh one otherwise"""
    B, nt = z_feed.shape
    n = z_core.shape[1]
    dev = z_feed.device
    o = OutsideSelection()
    o.B, o.nt, o.cap = B, nt, B * nt
    f32, i32 = torch.float32, torch.int32
    o.pts4 = torch.empty(o.cap, 4, dtype=f32, device=dev)
    o.dirs = torch.empty(o.cap, 3, dtype=f32, device=dev)
    o.dists = torch.empty(o.cap, dtype=f32, device=dev)
    o.sel = torch.empty(o.cap, dtype=i32, device=dev)
    o.count = torch.empty(1, dtype=i32, device=dev) if count is None else count
    o.alpha_full = torch.empty(B, nt, dtype=f32, device=dev)
    o.rgb_full = torch.empty(B, nt, 3, dtype=f32, device=dev)
    work = torch.empty(B + o.cap, dtype=i32, device=dev)
    _launch("fneus_outside_select", lib.fneus_outside_select, _ptr(rays_o), _ptr(rays_d), _ptr(z_core), _ptr(z_feed), B, n, nt,
            float(sample_dist), _ptr(work), _ptr(o.pts4), _ptr(o.dirs), _ptr(o.dists), _ptr(o.sel), _ptr(o.count), _ptr(o.alpha_full),
            _ptr(o.rgb_full), _stream())
    return o


def outside_alpha_sel_fwd(density, rgb_raw, s: OutsideSelection):
    """alpha / colour of the listed samples into s.alpha_full / s.rgb_full (zeros elsewhere)"""
    _launch("fneus_outside_alpha_sel_fwd", lib.fneus_outside_alpha_sel_fwd, _ptr(density), _ptr(rgb_raw), _ptr(s.dists), _ptr(s.sel),
            _ptr(s.count), s.cap, _ptr(s.alpha_full), _ptr(s.rgb_full), _stream())
    return s.alpha_full, s.rgb_full


def outside_alpha_sel_bwd(density, rgb_full, dists, sel, count, d_alpha_full, d_rgb_full):
    cap = int(density.numel())
    d_density = torch.empty(cap, dtype=torch.float32, device=density.device)
    d_raw = torch.empty(cap, 3, dtype=torch.float32, device=density.device)
    _launch("fneus_outside_alpha_sel_bwd", lib.fneus_outside_alpha_sel_bwd, _ptr(density), _ptr(rgb_full), _ptr(dists), _ptr(sel),
            _ptr(count), cap, _ptr(d_alpha_full), _ptr(d_rgb_full), _ptr(d_density), _ptr(d_raw), _stream())
    return d_density, d_raw


def outside_alpha_fwd(density, rgb_raw, dists):
    n = int(density.numel())
    alpha = torch.empty(n, dtype=torch.float32, device=density.device)
    rgb = torch.empty(n, 3, dtype=torch.float32, device=density.device)
    _launch("fneus_outside_alpha_fwd", lib.fneus_outside_alpha_fwd, _ptr(density), _ptr(rgb_raw), _ptr(dists), n, _ptr(alpha),
            _ptr(rgb), _stream())
    return alpha, rgb


def outside_alpha_bwd(density, rgb, dists, d_alpha, d_rgb):
    n = int(density.numel())
    d_density = torch.empty(n, dtype=torch.float32, device=density.device)
    d_raw = torch.empty(n, 3, dtype=torch.float32, device=density.device)
    _launch("fneus_outside_alpha_bwd", lib.fneus_outside_alpha_bwd, _ptr(density), _ptr(rgb), _ptr(dists), _ptr(d_alpha), _ptr(d_rgb),
            n, _ptr(d_density), _ptr(d_raw), _stream())
    return d_density, d_raw


def stage2_loss(gt_lvis, pre_lvis, gt_rad, pre_rad, hit):
    """-> out [3] = (lvis_loss, radiance_loss, n_hit), d_pre_lvis, d_pre_rad (include/fneus.h fneus_stage2_loss)"""
    out = torch.empty(3, dtype=torch.float32, device=pre_lvis.device)
    d_l, d_r = torch.empty_like(pre_lvis), torch.empty_like(pre_rad)
    _launch("fneus_stage2_loss", lib.fneus_stage2_loss, _ptr(gt_lvis), _ptr(pre_lvis), _ptr(gt_rad), _ptr(pre_rad), _ptr(hit),
            pre_lvis.shape[0], _ptr(out), _ptr(d_l), _ptr(d_r), _stream())
    return out, d_l, d_r


def stage3_loss(rgb, true_rgb, mask, hit):
    """-> out [3] = (rgb_loss, psnr, sum of the weights), d_rgb [n, 3] = d rgb_loss / d rgb (include/fneus.h fneus_stage3_loss)"""
    out = torch.empty(3, dtype=torch.float32, device=rgb.device)
    d_rgb = torch.empty_like(rgb)
    _launch("fneus_stage3_loss", lib.fneus_stage3_loss, _ptr(rgb), _ptr(true_rgb), _ptr(mask), _ptr(hit), rgb.shape[0], _ptr(out),
            _ptr(d_rgb), _stream())
    return out, d_rgb


def latent_kl_fwd(latent, point_mask, rho: float, activated: bool = False):
    """-> stats [34]: rho_hat [32], the number of marked points, kl (include/fneus.h fneus_latent_kl_fwd); activated: `latent` is
    sigmoid(latent) already"""
    stats = torch.empty(34, dtype=torch.float32, device=latent.device)
    _launch("fneus_latent_kl_fwd", lib.fneus_latent_kl_fwd, _ptr(latent), _ptr(point_mask), latent.shape[0], float(rho), int(activated),
            _ptr(stats), _stream())
    return stats


def latent_kl_bwd(latent, point_mask, rho: float, stats, d_kl, activated: bool = False):
    d_latent = torch.empty_like(latent)
    _launch("fneus_latent_kl_bwd", lib.fneus_latent_kl_bwd, _ptr(latent), _ptr(point_mask), latent.shape[0], float(rho), int(activated),
            _ptr(stats), _ptr(d_kl), _ptr(d_latent), _stream())
    return d_latent


def sg_combine_fwd(sums, has_indir: bool):
    """[n, 4, 3] lobe sums of sg_render_fwd -> tone-mapped colour [n, 3] (include/fneus.h fneus_sg_combine_fwd)"""
    n = sums.shape[0]
    rgb = torch.empty(n, 3, dtype=torch.float32, device=sums.device)
    _launch("fneus_sg_combine_fwd", lib.fneus_sg_combine_fwd, _ptr(sums), n, int(has_indir), _ptr(rgb), _stream())
    return rgb


def sg_combine_bwd(sums, d_rgb, has_indir: bool):
    d_sums = torch.empty_like(sums)
    _launch("fneus_sg_combine_bwd", lib.fneus_sg_combine_bwd, _ptr(sums), _ptr(d_rgb), sums.shape[0], int(has_indir), _ptr(d_sums),
            _stream())
    return d_sums


def sg_render_fwd(lgt, ind, vis, normal, view, mat, f0: float):
    """render_with_sg's lobe sums (inverRender.py:314-449) -> [n, 4, 3]: direct specular / diffuse, indirect specular / diffuse"""
    n, M = mat.shape[0], lgt.shape[0]
    L = 0 if ind is None else ind.shape[1]
    out = torch.empty(n, 4, 3, dtype=torch.float32, device=mat.device)
    _launch("fneus_sg_render_fwd", lib.fneus_sg_render_fwd, _ptr(lgt), _ptr(ind), _ptr(vis), _ptr(normal), _ptr(view), _ptr(mat), n, M, L,
            float(f0), _ptr(out), _stream())
    return out


def sg_render_bwd(lgt, ind, vis, normal, view, mat, f0: float, d_out, d_lgt=None):
    """-> d_mat [n,7], d_lgt [M,7].  d_lgt given: the light table's gradient is ACCUMULATED into it (a persistent, cleared buffer)"""
    n, M = mat.shape[0], lgt.shape[0]
    L = 0 if ind is None else ind.shape[1]
    d_mat = torch.empty(n, 7, dtype=torch.float32, device=mat.device)
    if d_lgt is None:
        d_lgt = torch.zeros(M, 7, dtype=torch.float32, device=mat.device)
    _launch("fneus_sg_render_bwd", lib.fneus_sg_render_bwd, _ptr(lgt), _ptr(ind), _ptr(vis), _ptr(normal), _ptr(view), _ptr(mat), n, M, L,
            float(f0), _ptr(d_out), _ptr(d_mat), _ptr(d_lgt), _stream())
    return d_mat, d_lgt


def sg_render_heads_fwd(lgt, ind, vis, normal, view, brdf, cs, f0: float):
    """sg_render_fwd with the material taken from the two MLP heads' outputs: brdf [n,4] (diffuse albedo, raw roughness), cs [n,1]"""
    n, M = brdf.shape[0], lgt.shape[0]
    L = 0 if ind is None else ind.shape[1]
    out = torch.empty(n, 4, 3, dtype=torch.float32, device=brdf.device)
    _launch("fneus_sg_render_heads_fwd", lib.fneus_sg_render_heads_fwd, _ptr(lgt), _ptr(ind), _ptr(vis), _ptr(normal), _ptr(view), _ptr(brdf),
            _ptr(cs), n, M, L, float(f0), _ptr(out), _stream())
    return out


def sg_render_heads_bwd(lgt, ind, vis, normal, view, brdf, cs, f0: float, d_out, d_lgt=None):
    """-> d_brdf [n,4], d_cs [n,1], d_lgt [M,7] (accumulated into `d_lgt` when given)"""
    n, M = brdf.shape[0], lgt.shape[0]
    L = 0 if ind is None else ind.shape[1]
    d_brdf = torch.empty(n, 4, dtype=torch.float32, device=brdf.device)
    d_cs = torch.empty(n, 1, dtype=torch.float32, device=brdf.device)
    if d_lgt is None:
        d_lgt = torch.zeros(M, 7, dtype=torch.float32, device=brdf.device)
    _launch("fneus_sg_render_heads_bwd", lib.fneus_sg_render_heads_bwd, _ptr(lgt), _ptr(ind), _ptr(vis), _ptr(normal), _ptr(view), _ptr(brdf),
            _ptr(cs), n, M, L, float(f0), _ptr(d_out), _ptr(d_brdf), _ptr(d_cs), _ptr(d_lgt), _stream())
    return d_brdf, d_cs, d_lgt


def embed(x, n_freqs: int):
    """Embedder.embed (embedder.py:23-36) of a constant [n, d] input in one launch"""
    n, d = x.shape
    out = torch.empty(n, d * (1 + 2 * n_freqs), dtype=torch.float32, device=x.device)
    _launch("fneus_embed", lib.fneus_embed, _ptr(x), n, d, int(n_freqs), _ptr(out), _stream())
    return out


def lvis_h16_pack(blob):
    """the Lvis blob with every forward weight as one fp16 value (fneus_lvis_h16_pack): what fneus_lvis_visibility takes at prec 2"""
    out = torch.empty(int(lib.fneus_lvis_blob_bytes()), dtype=torch.uint8, device=blob.device)
    assert blob.numel() >= out.numel()
    _launch("fneus_lvis_h16_pack", lib.fneus_lvis_h16_pack, _ptr(blob), _ptr(out), _stream())
    return out


def lvis_visibility(blob, points, normals, dirs, weights, prec: int, point_mask=None):
    """get_diffuse_visibility's network part (inverRender.py:163-190) -> vis [M, n]; dirs [M, 32, 3], weights [M, 32];
    point_mask [n] bool / uint8: points marked 0 are skipped (their visibility is 0)"""
    n, M, S = points.shape[0], dirs.shape[0], dirs.shape[1]
    for t, name in ((points, "points"), (normals, "normals"), (dirs, "dirs"), (weights, "weights")):
        _chk_f32(t, name)
    vis = torch.empty(M, n, dtype=torch.float32, device=points.device)
    if point_mask is not None:
        point_mask = (point_mask.view(torch.uint8) if point_mask.dtype == torch.bool else point_mask.to(torch.uint8)).contiguous()
    _launch("fneus_lvis_visibility", lib.fneus_lvis_visibility, _ptr(blob), _ptr(points), _ptr(normals), _ptr(dirs), _ptr(weights),
            _ptr(point_mask), n, M, S, _ptr(vis), prec, _stream())
    return vis


def gen_random_rays(intrinsics_inv, pose, image, mask, px, py):
    """Dataset.gen_random_rays_at (dataset.py:133-151) for integer pixels px, py (int64 [n]) of one image -> [n,10]"""
    n = int(px.numel())
    H, W = int(image.shape[0]), int(image.shape[1])
    for t, name in ((intrinsics_inv, "intrinsics_inv"), (pose, "pose"), (image, "image"), (mask, "mask")):
        _chk_f32(t, name)
    if px.dtype != torch.int64 or py.dtype != torch.int64 or not px.is_cuda or not py.is_cuda:
        raise ValueError("pixel coordinates must be int64 CUDA tensors")
    out = torch.empty(n, 10, dtype=torch.float32, device=image.device)
    _launch("fneus_gen_random_rays", lib.fneus_gen_random_rays, _ptr(intrinsics_inv), _ptr(pose), _ptr(image), _ptr(mask), H, W,
            _ptr(px.contiguous()), _ptr(py.contiguous()), n, _ptr(out), _stream())
    return out


def gen_rays_grid(intrinsics_inv, pose, tx, ty):
    """Dataset.gen_rays_at (dataset.py:115-131) at pixel positions tx [nx] x ty [ny] -> rays_o, rays_v [ny, nx, 3]"""
    nx, ny = int(tx.numel()), int(ty.numel())
    for t, name in ((intrinsics_inv, "intrinsics_inv"), (pose, "pose"), (tx, "tx"), (ty, "ty")):
        _chk_f32(t, name)
    o = torch.empty(ny, nx, 3, dtype=torch.float32, device=tx.device)
    v = torch.empty(ny, nx, 3, dtype=torch.float32, device=tx.device)
    _launch("fneus_gen_rays_grid", lib.fneus_gen_rays_grid, _ptr(intrinsics_inv), _ptr(pose), _ptr(tx), _ptr(ty), nx, ny, _ptr(o),
            _ptr(v), _stream())
    return o, v


def ray_hit(rays_o, rays_d, mid_z, sdf, inside_mask=None, dists=None, normal=None, inv_s: float = 1.0, want_weights=False):
    """first surface hit per ray (+ the occlusion of compute_weight when normal / dists are given), fneus_ray_hit.
    -> dict(sdf_mask u8 [B], z_surf [B], pts_surf [B,3][, occlusion [B]][, weights [B,n]])"""
    B, n = mid_z.shape
    dev = mid_z.device
    out = {"sdf_mask": torch.empty(B, dtype=torch.uint8, device=dev), "z_surf": torch.empty(B, dtype=torch.float32, device=dev),
           "pts_surf": torch.empty(B, 3, dtype=torch.float32, device=dev)}
    if normal is not None:
        out["occlusion"] = torch.empty(B, dtype=torch.float32, device=dev)
        if want_weights:
            out["weights"] = torch.empty(B, n, dtype=torch.float32, device=dev)
    if inside_mask is not None:
        inside_mask = inside_mask.to(torch.uint8).contiguous()
    _launch("fneus_ray_hit", lib.fneus_ray_hit, _ptr(rays_o), _ptr(rays_d), _ptr(mid_z), _ptr(sdf), _ptr(dists), _ptr(normal),
            _ptr(inside_mask), B, n, float(inv_s), _ptr(out["sdf_mask"]), _ptr(out["z_surf"]), _ptr(out["pts_surf"]),
            _ptr(out.get("occlusion")), _ptr(out.get("weights")), _stream())
    return out


def sample_dirs(surf, normal, u_theta, u_z):
    """calLvis.py:302-320 + :351-357 -> origins [M*S,3], dirs [M*S,3]"""
    M, S = u_theta.shape
    origins = torch.empty(M * S, 3, dtype=torch.float32, device=surf.device)
    dirs = torch.empty(M * S, 3, dtype=torch.float32, device=surf.device)
    _launch("fneus_sample_dirs", lib.fneus_sample_dirs, _ptr(surf), _ptr(normal), _ptr(u_theta), _ptr(u_z), M, S, _ptr(origins),
            _ptr(dirs), _stream())
    return origins, dirs


def merge(z_old, s_old, z_new, s_new):
    B, m = z_old.shape
    k = z_new.shape[1]
    z_out = torch.empty(B, m + k, dtype=torch.float32, device=z_old.device)
    s_out = torch.empty_like(z_out) if s_new is not None else None
    _launch("fneus_merge", lib.fneus_merge, _ptr(z_old), _ptr(s_old) if s_new is not None else None, m, _ptr(z_new), _ptr(s_new), k, B,
                          _ptr(z_out), _ptr(s_out), _stream())
    return z_out, s_out


def merge_upsample(rays_o, rays_d, z_old, s_old, z_new, s_new, inv_s: float, k_next: int, last: bool, sample_dist=None):
    """cat_z_vals of one up-sampling step + up_sample of the next in one launch -> z_out, s_out [B, m + k], z_next [B, k_next],
    z_final [B, m + k + k_next] (last step only, else None).  sample_dist given (with last): also the sections of z_final in
    the same launch; the result is then (z_out, s_out, z_next, z_final, dists, mid_z)"""
    B, m = z_old.shape
    k = z_new.shape[1]
    dev = z_old.device
    z_out = torch.empty(B, m + k, dtype=torch.float32, device=dev)
    s_out = torch.empty_like(z_out)
    z_next = torch.empty(B, k_next, dtype=torch.float32, device=dev)
    z_final = torch.empty(B, m + k + k_next, dtype=torch.float32, device=dev) if last else None
    sec = last and sample_dist is not None
    dists = torch.empty_like(z_final) if sec else None
    mid_z = torch.empty_like(z_final) if sec else None
    _launch("fneus_merge_upsample", lib.fneus_merge_upsample, _ptr(rays_o), _ptr(rays_d), _ptr(z_old), _ptr(s_old), m, _ptr(z_new),
            _ptr(s_new), k, B, float(inv_s), int(k_next), _ptr(z_out), _ptr(s_out), _ptr(z_next), _ptr(z_final),
            float(sample_dist) if sec else 0.0, _ptr(dists), _ptr(mid_z), _stream())
    if sample_dist is not None:
        return z_out, s_out, z_next, z_final, dists, mid_z
    return z_out, s_out, z_next, z_final


SAMPLER_K1_FUSED = _os.environ.get("FNEUS_SAMPLER_K1_FUSED", "1") != "0"


def sdf_merge_upsample(blob, prec: int, rays_o, rays_d, z_old, s_old, z_new, inv_s: float, k_next: int, last: bool, sample_dist=None,
                       want_s_new: bool = False):
    """fneus_sdf_fwd on the new depths z_new [B, k] of an up-sampling step + merge_upsample(...) of that step in ONE launch
    (fneus_sdf_fwd_merge_upsample; bit-identical to the two).  -> merge_upsample's result (+ s_new [B, k] with want_s_new), or None when
    the launch does not take the shape (the caller then runs the two launches)."""
    B, m = z_old.shape
    k = z_new.shape[1]
    if not SAMPLER_K1_FUSED or PROFILE is not None or k not in (16, 32) or (B * k + 31) // 32 >= 1024 or m + k + k_next > 256:
        return None
    dev = z_old.device
    z_out = torch.empty(B, m + k, dtype=torch.float32, device=dev)
    s_out = torch.empty_like(z_out)
    z_next = torch.empty(B, k_next, dtype=torch.float32, device=dev)
    z_final = torch.empty(B, m + k + k_next, dtype=torch.float32, device=dev) if last else None
    sec = last and sample_dist is not None
    dists = torch.empty_like(z_final) if sec else None
    mid_z = torch.empty_like(z_final) if sec else None
    s_new = torch.empty(B, k, dtype=torch.float32, device=dev) if want_s_new else None
    _launch("fneus_sdf_fwd_merge_upsample", lib.fneus_sdf_fwd_merge_upsample, _ptr(blob), _ptr(rays_o), _ptr(rays_d), _ptr(z_old), _ptr(s_old), m,
            _ptr(z_new), k, B, float(inv_s), int(k_next), _ptr(z_out), _ptr(s_out), _ptr(z_next), _ptr(z_final),
            float(sample_dist) if sec else 0.0, _ptr(dists), _ptr(mid_z), _ptr(s_new), prec, _stream())
    out = (z_out, s_out, z_next, z_final, dists, mid_z) if sample_dist is not None else (z_out, s_out, z_next, z_final)
    return out + (s_new,) if want_s_new else out


SAMPLER_STEPS_FUSED = _os.environ.get("FNEUS_SAMPLER_STEPS_FUSED", "1") != "0"


def sdf_merge_upsample_steps(blob, prec: int, rays_o, rays_d, z_old, s_old, z_new, inv_s_list, k_next: int, sample_dist: float):
    """every remaining step of the hierarchical sampler in ONE launch (fneus_sdf_fwd_merge_upsample_steps): step j evaluates its new depths,
    merges them and draws the next ones; inv_s_list = the steps' 64 * 2^i.  -> (z_final, dists, mid_z), or None when the launch does
    not take the shape (the caller then runs the steps one by one)."""
    B, m = z_old.shape
    k = z_new.shape[1]
    n = len(inv_s_list)
    if (not SAMPLER_STEPS_FUSED or not SAMPLER_K1_FUSED or PROFILE is not None or n < 2 or n > 4 or k not in (16, 32) or k_next != k
            or (B * k + 31) // 32 >= 1024 or m + k * (n + 1) > 256):
        return None
    dev = z_old.device
    f32 = dict(dtype=torch.float32, device=dev)
    steps = (_lib.FneusSamplerStep * n)()
    keep = []
    zo, so, zn = z_old, s_old, z_new
    for j in range(n):
        mj = m + j * k
        z_out, s_out, z_next = torch.empty(B, mj + k, **f32), torch.empty(B, mj + k, **f32), torch.empty(B, k_next, **f32)
        st = steps[j]
        st.z_old, st.s_old, st.m, st.z_new, st.inv_s = zo.data_ptr(), so.data_ptr(), mj, zn.data_ptr(), float(inv_s_list[j])
        st.z_out, st.s_out, st.z_next = z_out.data_ptr(), s_out.data_ptr(), z_next.data_ptr()
        keep += [z_out, s_out, z_next]
        zo, so, zn = z_out, s_out, z_next
    z_final = torch.empty(B, m + n * k + k_next, **f32)
    dists, mid_z = torch.empty_like(z_final), torch.empty_like(z_final)
    _launch("fneus_sdf_fwd_merge_upsample", lib.fneus_sdf_fwd_merge_upsample_steps, _ptr(blob), _ptr(rays_o), _ptr(rays_d), n, steps, k, B,
            int(k_next), _ptr(z_final), float(sample_dist), _ptr(dists), _ptr(mid_z), prec, _stream())
    z_final._chain = keep              # (the intermediate arrays live as long as the result: the launch may still be queued)
    return z_final, dists, mid_z


def split_batch(data: torch.Tensor):
    """[B,10] batch -> contiguous rays_o [B,3], rays_d [B,3], rgb [B,3], mask [B,1] in one launch"""
    _chk_f32(data, "data")
    B = data.shape[0]
    f32 = dict(dtype=torch.float32, device=data.device)
    o, d, rgb, mask = torch.empty(B, 3, **f32), torch.empty(B, 3, **f32), torch.empty(B, 3, **f32), torch.empty(B, 1, **f32)
    _launch("fneus_split_batch", lib.fneus_split_batch, _ptr(data), B, _ptr(o), _ptr(d), _ptr(rgb), _ptr(mask), _stream())
    return o, d, rgb, mask


def ray_setup(rays_o, rays_d, n_samples: int, near=None, far=None, t_rand=None):
    """coarse depths z_vals [B, n_samples]; near / far None -> unit-sphere bounds computed from the rays"""
    B = rays_o.shape[0]
    z = torch.empty(B, n_samples, dtype=torch.float32, device=rays_o.device)
    _launch("fneus_ray_setup", lib.fneus_ray_setup, _ptr(rays_o), _ptr(rays_d), _ptr(near), _ptr(far), _ptr(t_rand), B, n_samples,
            _ptr(z), _stream())
    return z


def sections(z, sample_dist: float):
    B, n = z.shape
    dists = torch.empty_like(z)
    mid_z = torch.empty_like(z)
    _launch("fneus_sections", lib.fneus_sections, _ptr(z), B, n, float(sample_dist), _ptr(dists), _ptr(mid_z), _stream())
    return dists, mid_z


def _car(car):
    """cos_anneal_ratio as (float, device pointer): a 1-element device tensor is read by the kernel at run time (so that
    a captured step can be replayed with a new value), a Python number is passed by value"""
    if torch.is_tensor(car):
        _chk_f32(car, "cos_anneal_ratio")
        return 0.0, _ptr(car)
    return float(car), None


def _back_rows(back_rgb, B):
    if back_rgb is None:
        return None, 0
    back_rgb = back_rgb.detach().float().reshape(-1, 3).contiguous()
    assert back_rgb.shape[0] in (1, B), "background_rgb: one colour or one per ray"
    return back_rgb, int(back_rgb.shape[0])


def composite_fwd(rays_o, rays_d, mid_z, dists, sdf, normal, rgb, inv_s, car: float, bg_alpha=None, bg_color=None,
                  inv_s_mode: int = 0, back_rgb=None):
    """inv_s_mode 1: `inv_s` is the variance parameter (see include/fneus.h).  out["eik"] is [2, B].
    back_rgb [1, 3] / [B, 3]: the constant background colour of renderer.py:367-368, added inside the kernel"""
    B, n = mid_z.shape
    n_out = 0 if bg_alpha is None else bg_alpha.shape[1] - n
    nt = n + n_out
    dev = mid_z.device
    back, back_rows = _back_rows(back_rgb, B)
    f32 = dict(dtype=torch.float32, device=dev)
    out = {
        "weights": torch.empty(B, nt, **f32), "color": torch.empty(B, 3, **f32), "wsum": torch.empty(B, **f32),
        "wmax": torch.empty(B, **f32), "cdf": torch.empty(B, n, **f32), "inside": torch.empty(B, n, **f32),
        "eik": torch.empty(2, B, **f32), "min_idx": torch.empty(B, dtype=torch.int32, device=dev),
        "sdf_mask": torch.empty(B, dtype=torch.uint8, device=dev), "wpair": torch.empty(B, 2, **f32),
    }
    _launch("fneus_composite_fwd", lib.fneus_composite_fwd, _ptr(rays_o), _ptr(rays_d), _ptr(mid_z), _ptr(dists), _ptr(sdf),
            _ptr(normal), _ptr(rgb), _ptr(inv_s), int(inv_s_mode), B, n, *_car(car), _ptr(bg_alpha), _ptr(bg_color), n_out,
            _ptr(out["weights"]), _ptr(out["color"]), _ptr(out["wsum"]), _ptr(out["wmax"]), _ptr(out["cdf"]),
            _ptr(out["inside"]), _ptr(out["eik"]), _ptr(out["min_idx"]), _ptr(out["sdf_mask"]), _ptr(out["wpair"]),
            _ptr(back), back_rows, _stream())
    return out


def composite_bwd(rays_o, rays_d, mid_z, dists, sdf, normal, rgb, inv_s, car, min_idx, sdf_mask, d_color, d_wsum,
                  d_weights, d_wpair, d_eiknum, bg_alpha=None, bg_color=None, inv_s_mode: int = 0, back_rgb=None):
    B, n = mid_z.shape
    back, back_rows = _back_rows(back_rgb, B)
    n_out = 0 if bg_alpha is None else bg_alpha.shape[1] - n
    dev = mid_z.device
    d_sdf = torch.empty(B * n, dtype=torch.float32, device=dev)
    d_normal = torch.empty(B * n, 3, dtype=torch.float32, device=dev)
    d_rgb = torch.empty(B * n, 3, dtype=torch.float32, device=dev)
    d_inv_s = torch.empty(B, dtype=torch.float32, device=dev)
    d_bga = torch.empty_like(bg_alpha) if bg_alpha is not None else None
    d_bgc = torch.empty_like(bg_color) if bg_color is not None else None
    _launch("fneus_composite_bwd", lib.fneus_composite_bwd, _ptr(rays_o), _ptr(rays_d), _ptr(mid_z), _ptr(dists), _ptr(sdf),
            _ptr(normal), _ptr(rgb), _ptr(inv_s), int(inv_s_mode), B, n, *_car(car), _ptr(bg_alpha), _ptr(bg_color), n_out, _ptr(min_idx),
            _ptr(sdf_mask), _ptr(d_color), _ptr(d_wsum), _ptr(d_weights), _ptr(d_wpair), _ptr(d_eiknum), _ptr(d_sdf),
            _ptr(d_normal), _ptr(d_rgb), _ptr(d_inv_s), _ptr(d_bga), _ptr(d_bgc), _ptr(back), back_rows, _stream())
    return d_sdf, d_normal, d_rgb, d_inv_s, d_bga, d_bgc


# ------------------------------------------------------------------------------------------------------------
# box calibration (bench.py `box`)
# ------------------------------------------------------------------------------------------------------------
def box_probe(device, mfma_iters: int = 100, copy_mib: int = 512, reps: int = 5):
    """two ~100 us probes of THIS box: back-to-back bf16 MFMAs on random operands (TFLOP/s and the clock the chip held) and a float4
    copy (TB/s, bytes read + written) -- fneus_probe_mfma / fneus_probe_copy (csrc/probe.hip)"""
    g = torch.Generator(device=device).manual_seed(7)
    operands = (torch.rand(16384, device=device, generator=g) * 2 - 1).to(torch.bfloat16)
    ticks = torch.zeros(2, dtype=torch.int64, device=device)
    sink = torch.zeros(1, dtype=torch.float32, device=device)
    src = torch.rand(copy_mib << 18, device=device, generator=g)          # copy_mib MiB of fp32
    dst = torch.empty_like(src)

    def timed(fn):
        fn()
        best = 1e30
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        return best

    t_m = timed(lambda: check(lib.fneus_probe_mfma(_ptr(operands), mfma_iters, _ptr(ticks), _ptr(sink), _stream()), "fneus_probe_mfma"))
    cyc, rt = [int(v) for v in ticks.cpu()]
    t_c = timed(lambda: check(lib.fneus_probe_copy(_ptr(src), _ptr(dst), src.numel() * 4, _stream()), "fneus_probe_copy"))
    return {"mfma_bf16_tflops": float(lib.fneus_probe_mfma_flops(mfma_iters)) / t_m / 1e12,
            "mfma_clock_ghz": (cyc / rt * 0.1) if rt > 0 else None,
            "copy_tbs": 2.0 * src.numel() * 4 / t_c / 1e12,
            "what": f"best of {reps}: {mfma_iters} x 16 back-to-back v_mfma_f32_32x32x16_bf16 per wave on random operands, 16 waves per CU "
                    f"(TFLOP/s by HIP events, clock = shader cycles / 100 MHz ticks inside the loop); float4 copy of {copy_mib} MiB (read + written bytes)"}
