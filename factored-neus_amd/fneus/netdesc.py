"""Host-side description of the two packed MLPs: flat parameter layouts, index maps and pack-job tables.

The device packer (csrc/pack.hip) is generic; everything network specific (skip connection, the 1/sqrt(2) fold,
the row order of the last SDF layer, the input order of the colour network) lives in the index maps built here,
once per process.  Geometry and blob offsets come from the library itself (fneus_layout), not from duplicated
constants.

Reference: SDFNetwork (models/fields.py:9-91), RenderingNetwork (models/fields.py:114-175).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List

import numpy as np

from . import _lib

PACK_FRAG, PACK_ACCVEC = 0, 1
JOB_DTYPE = np.dtype([("kind", "<i4"), ("unit_base", "<i4"), ("dst_hi", "<u4"), ("dst_lo", "<u4"), ("src", "<u4"),
                      ("ld", "<i4"), ("ks", "<i4"), ("nt", "<i4"), ("transposed", "<i4"), ("rowmap", "<u4"),
                      ("kmap", "<u4"), ("scale", "<f4"), ("rs_base", "<i4"), ("rs_mode", "<i4"), ("geom", "<i4"),
                      ("pad", "<i4")])
ROW_DTYPE = np.dtype([("off_v", "<u4"), ("off_g", "<u4"), ("n_in", "<i4"), ("off_w_eff", "<u4")])

SDF_IN = [39, 256, 256, 256, 256, 256, 256, 256, 256]
SDF_OUT = [256, 256, 256, 217, 256, 256, 256, 256, 257]
COL_IN = [289, 256, 256, 256, 256]
COL_OUT = [256, 256, 256, 256, 3]
N_PE = 39          # SDF positional encoding width (multires 6)
N_SIDE = 33        # colour-network inputs besides the feature: pts 3 + PE4(view) 27 + normal 3


def phi(ks, h, j):
    return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)


def slot_features(ks_count):
    """feature index held by k-slot kappa = 16*ks + 8*h + j"""
    out = np.empty(ks_count * 16, dtype=np.int64)
    for ks in range(ks_count):
        for h in range(2):
            for j in range(8):
                out[16 * ks + 8 * h + j] = phi(ks, h, j)
    return out


@dataclass
class Layout:
    n_layers: int
    total: int
    extra: int
    off: np.ndarray      # [L, 5] fwd_hi, fwd_lo, rev_hi, rev_lo, bias
    geom: np.ndarray     # [L, 4] ksf, ntf, ksr, ntr


def query_layout(which: int) -> Layout:
    buf = (C.c_int32 * 256)()
    n = _lib.lib.fneus_layout(which, buf, 256)
    if n <= 0:
        raise RuntimeError("fneus_layout failed")
    a = np.frombuffer(buf, dtype=np.int32, count=n).copy()
    L = int(a[0])
    per = a[3:3 + 9 * L].reshape(L, 9)
    return Layout(L, int(a[1]), int(a[2]), per[:, :5].copy(), per[:, 5:].copy())


NO_G = 0xFFFFFFFF      # RowInfo.off_g of a plain nn.Linear row (csrc/fneus_pack.h)


def raw_offsets(ins, outs, weight_norm=True):
    """Raw (module) parameter layout per layer: bias[out], weight_g[out] (weight-normalised layers only),
    weight_v / weight [out][in]."""
    offB, offG, offV, o = [], [], [], 0
    for i, k in zip(ins, outs):
        offB.append(o)
        o += k
        if weight_norm:
            offG.append(o)
            o += k
        else:
            offG.append(None)
        offV.append(o)
        o += i * k
    return offB, offG, offV, o


def row_table(ins, outs, offW_eff, weight_norm=True):
    """RowInfo per output row + the first row index of each layer"""
    offB, offG, offV, _ = raw_offsets(ins, outs, weight_norm)
    rows, base = [], []
    for l, (i, k) in enumerate(zip(ins, outs)):
        base.append(len(rows))
        for r in range(k):
            rows.append((offV[l] + r * i, offG[l] + r if weight_norm else NO_G, i, offW_eff[l] + r * i))
    return np.array(rows, dtype=ROW_DTYPE), base


def flat_offsets(ins, outs):
    """(offW, offb, total) of the flat fp32 parameter buffer: per layer W[out][in] then b[out]."""
    offW, offb, o = [], [], 0
    for i, k in zip(ins, outs):
        offW.append(o)
        o += i * k
        offb.append(o)
        o += k
    return offW, offb, o


class _Builder:
    def __init__(self, geom=0):
        self.geom = geom                       # 0: 32-row tiles / 16-deep k-steps, 1: 16-row tiles / 32-deep k-steps
        self.tr, self.kd = (16, 32) if geom else (32, 16)
        self.maps: List[np.ndarray] = []
        self.n_map = 0
        self.jobs = []
        self.units = 0

    def add_map(self, arr):
        arr = np.asarray(arr, dtype=np.int32)
        off = self.n_map
        self.maps.append(arr)
        self.n_map += arr.size
        return off

    def frag(self, dst_hi, dst_lo, src, ld, ks, nt, transposed, rowmap, kmap, scale=1.0, rs_base=-1):
        assert len(rowmap) == nt * self.tr and len(kmap) == ks * self.kd
        self.jobs.append((PACK_FRAG, self.units, dst_hi, dst_lo, src, ld, ks, nt, transposed,
                          self.add_map(rowmap), self.add_map(kmap), scale, rs_base, 1 if transposed else 0, self.geom, 0))
        self.units += ks * nt

    def accvec(self, dst, src, ld, nt, rowmap, scale=1.0, rs_base=-1, rs_mode=0):
        assert len(rowmap) == nt * self.tr
        self.jobs.append((PACK_ACCVEC, self.units, dst, 0, src, ld, 0, nt, 0, self.add_map(rowmap), 0, scale,
                          rs_base, rs_mode, self.geom, 0))
        self.units += nt

    def finish(self):
        jobs = np.array(self.jobs, dtype=JOB_DTYPE)
        maps = np.concatenate(self.maps).astype(np.int32)
        return jobs, maps, self.units


def _lim(idx, n):
    idx = np.asarray(idx, dtype=np.int64)
    return np.where(idx < n, idx, -1)


def build_sdf_jobs():
    ly = query_layout(0)
    offW, offb, total = flat_offsets(SDF_IN, SDF_OUT)          # effective layout (gradients of W_eff, b)
    offB, offG, offV, total_raw = raw_offsets(SDF_IN, SDF_OUT)  # module parameters (bias, weight_g, weight_v)
    row_tab, rbase = row_table(SDF_IN, SDF_OUT, offW)
    b = _Builder()
    inv_sqrt2 = 1.0 / math.sqrt(2.0)
    for l in range(9):
        ksf, ntf, ksr, ntr = [int(v) for v in ly.geom[l]]
        fwd_hi, fwd_lo, rev_hi, rev_lo, bias = [int(v) for v in ly.off[l]]
        n_in, n_out = SDF_IN[l], SDF_OUT[l]
        scale = inv_sqrt2 if l == 4 else 1.0
        # ---- forward: rows = outputs, k-slots = inputs
        rows = np.arange(ntf * 32)
        if l == 8:
            rowmap = np.where(rows < 256, rows + 1, np.where(rows == 256, 0, -1))
        else:
            rowmap = _lim(rows, n_out)
        feat = slot_features(ksf)
        if l == 4:
            kmap = np.where(feat < 224, _lim(feat, 217), np.where(feat - 224 < N_PE, 217 + (feat - 224), -1))
        else:
            kmap = _lim(feat, n_in)
        b.frag(fwd_hi, fwd_lo, offV[l], n_in, ksf, ntf, 0, rowmap, kmap, scale, rbase[l])
        # ---- reverse (A = W^T): rows = inputs, k-slots = outputs (accumulator order of a_l / zbar_l)
        ofeat = slot_features(ksr)
        if l == 8:
            kmap_r = np.where(ofeat < 256, ofeat + 1, np.where(ofeat == 256, 0, -1))
        else:
            kmap_r = _lim(ofeat, n_out)
        rin = np.arange(ntr * 32)
        if l == 4:
            rowmap_r = np.where(rin < 224, _lim(rin, 217), np.where(rin - 224 < N_PE, 217 + (rin - 224), -1))
        else:
            rowmap_r = _lim(rin, n_in)
        b.frag(rev_hi, rev_lo, offV[l], n_in, ksr, ntr, 1, rowmap_r, kmap_r, scale, rbase[l])
        b.accvec(bias, offB[l], 1, ntf, rowmap)
    # row 0 of the last layer in accumulator layout: g_hat(h_8) of the reverse sweep
    b.accvec(ly.extra, offV[8], 1, 8, np.arange(256), rs_base=rbase[8], rs_mode=2)
    jobs, maps, units = b.finish()
    segs = np.array([(offb[l], offB[l], SDF_OUT[l], 0) for l in range(9)], dtype=np.int32)
    return {"layout": ly, "jobs": jobs, "maps": maps, "units": units, "n_params": total, "offW": offW, "offb": offb,
            "ins": SDF_IN, "outs": SDF_OUT, "n_raw": total_raw, "offB": offB, "offG": offG, "offV": offV,
            "rows": row_tab, "bias_segs": segs}


# the two MLPs of the surface head RefColor (reference models/fields.py:285-301): colour-network geometry, plain Linear
REFCD_IN, REFCD_OUT, REFCD_SIDE = [286, 256, 256, 256, 256], [256, 256, 256, 256, 3], 30    # net_cd: pts | PE4(n) | feat
REFVD_IN, REFVD_OUT, REFVD_SIDE = [289, 256, 256, 256, 256], [256, 256, 256, 256, 1], 33    # viewdir_mlp + net_cs


def build_color_jobs(COL_IN=COL_IN, COL_OUT=COL_OUT, N_SIDE=N_SIDE, weight_norm=True):
    """Pack tables of a colour-shaped MLP: layer 0 takes N_SIDE side inputs (reference columns 0..N_SIDE-1) followed by
    the 256 features; the kernels keep the features in the first 16 k-steps and the side inputs in the last 3."""
    ly = query_layout(1)
    offW, offb, total = flat_offsets(COL_IN, COL_OUT)
    offB, offG, offV, total_raw = raw_offsets(COL_IN, COL_OUT, weight_norm)
    row_tab, rbase = row_table(COL_IN, COL_OUT, offW, weight_norm)
    b = _Builder()
    for l in range(5):
        ksf, ntf, ksr, ntr = [int(v) for v in ly.geom[l]]
        fwd_hi, fwd_lo, rev_hi, rev_lo, bias = [int(v) for v in ly.off[l]]
        n_in, n_out = COL_IN[l], COL_OUT[l]
        rowmap = _lim(np.arange(ntf * 32), n_out)
        feat = slot_features(ksf)
        if l == 0:   # k-slots: 256 feature slots first, then the 33 side inputs (reference column order: side | feature)
            kmap = np.where(feat < 256, feat + N_SIDE, np.where(feat - 256 < N_SIDE, feat - 256, -1))
        else:
            kmap = _lim(feat, n_in)
        rs = rbase[l] if weight_norm else -1          # plain Linear layers: no row scale to apply
        b.frag(fwd_hi, fwd_lo, offV[l], n_in, ksf, ntf, 0, rowmap, kmap, 1.0, rs)
        kmap_r = _lim(slot_features(ksr), n_out)
        rin = np.arange(ntr * 32)
        if l == 0:
            rowmap_r = np.where(rin < 256, rin + N_SIDE, np.where(rin - 256 < N_SIDE, rin - 256, -1))
        else:
            rowmap_r = _lim(rin, n_in)
        b.frag(rev_hi, rev_lo, offV[l], n_in, ksr, ntr, 1, rowmap_r, kmap_r, 1.0, rs)
        b.accvec(bias, offB[l], 1, ntf, rowmap)
    # the (<= 3) rows of the last layer in accumulator layout: the two-pass forward kernel takes 256 -> 3 as vector dot products
    for r in range(min(COL_OUT[4], 3)):
        b.accvec(ly.extra + r * 8 * 2 * 16 * 4, offV[4] + r * COL_IN[4], 1, 8, np.arange(256),
                 rs_base=(rbase[4] + r) if weight_norm else -1, rs_mode=2)
    jobs, maps, units = b.finish()
    segs = np.array([(offb[l], offB[l], COL_OUT[l], 0) for l in range(5)], dtype=np.int32)
    return {"layout": ly, "jobs": jobs, "maps": maps, "units": units, "n_params": total, "offW": offW, "offb": offb,
            "ins": COL_IN, "outs": COL_OUT, "n_raw": total_raw, "offB": offB, "offG": offG, "offV": offV,
            "rows": row_tab, "bias_segs": segs, "n_side": N_SIDE, "weight_norm": weight_norm}


def build_refcd_jobs():
    return build_color_jobs(REFCD_IN, REFCD_OUT, REFCD_SIDE, weight_norm=False)


def build_refvd_jobs():
    return build_color_jobs(REFVD_IN, REFVD_OUT, REFVD_SIDE, weight_norm=False)



# ---- background NeRF++ (reference models/fields.py:178-259), plain nn.Linear layers ------------------------------------
# parameter layers in the order of NERF_NAMES; feature_linear and alpha_linear are adjacent in the flat buffer so that
# their weights form ONE [257][256] matrix (and their biases one [257] vector) for the shared pack entry 9
NERF_NAMES = [f"pts_linears.{i}" for i in range(8)] + ["feature_linear", "alpha_linear", "views_linears.0", "rgb_linear"]
NERF_IN = [84, 256, 256, 256, 256, 340, 256, 256, 256, 256, 283, 128]
NERF_OUT = [256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3]
NERF_PE, NERF_VIEW_PE = 84, 27


def nerf_raw_offsets():
    offB, offV, o = [], [], 0
    for l, (i, k) in enumerate(zip(NERF_IN, NERF_OUT)):
        if NERF_NAMES[l] == "feature_linear":       # b_feature[256] b_alpha[1] W_feature[256][256] W_alpha[1][256]
            offB += [o, o + 256]
            offV += [o + 257, o + 257 + 256 * 256]
            o += 257 + 257 * 256
        elif NERF_NAMES[l] == "alpha_linear":
            continue
        else:
            offB.append(o)
            o += k
            offV.append(o)
            o += i * k
    return offB, offV, o


def build_nerf_jobs():
    """Pack tables of the background NeRF++ (csrc/fneus_layout.h kNerfGeom: one entry per pack).  There is no separate
    effective-parameter buffer: the layers are plain Linear, the weight-gradient GEMM accumulates straight into the raw
    gradient (offW / offb = offV / offB)."""
    ly = query_layout(2)
    offB, offV, total_raw = nerf_raw_offsets()
    b = _Builder()
    P = {name: i for i, name in enumerate(NERF_NAMES)}
    # pack entry -> (parameter layer, forward k-map as reference column per feature slot index, reverse row map)
    for e in range(12):
        ksf, ntf, ksr, ntr = [int(v) for v in ly.geom[e]]
        fwd_hi, fwd_lo, rev_hi, rev_lo, bias = [int(v) for v in ly.off[e]]
        feat = slot_features(ksf)
        rows = np.arange(ntf * 32)
        rin = np.arange(ntr * 32)
        with_bias, n_rows_src = True, None
        if e == 0:
            pl, kmap, rowmap_r = P["pts_linears.0"], _lim(feat, NERF_PE), None
        elif e in (1, 2, 3, 4):
            pl, kmap, rowmap_r = e, _lim(feat, 256), _lim(rin, 256)
        elif e == 5:        # pts_linears.5, h columns: reference input = cat([input_pts 84, h 256]) (fields.py:245)
            pl, kmap, rowmap_r = 5, np.where(feat < 256, feat + NERF_PE, -1), np.where(rin < 256, rin + NERF_PE, -1)
        elif e == 6:        # pts_linears.5, encoding columns (accumulates into the same tiles; bias lives in entry 5)
            pl, kmap, rowmap_r, with_bias = 5, _lim(feat, NERF_PE), None, False
        elif e in (7, 8):
            pl, kmap, rowmap_r = e - 1, _lim(feat, 256), _lim(rin, 256)
        elif e == 9:        # rows 0..255 feature_linear, row 256 alpha_linear: one [257][256] matrix in the flat buffer
            pl, kmap, rowmap_r, n_rows_src = P["feature_linear"], _lim(feat, 256), _lim(rin, 256), 257
        elif e == 10:       # views_linears.0: columns [feature 256 | PE4(view) 27]; reverse only onto the feature inputs
            pl, kmap, rowmap_r = P["views_linears.0"], _lim(feat, 256 + NERF_VIEW_PE), _lim(rin, 256)
        else:
            pl, kmap, rowmap_r = P["rgb_linear"], _lim(feat, 128), _lim(rin, 128)
        n_in = NERF_IN[pl]
        n_out = n_rows_src if n_rows_src else NERF_OUT[pl]
        rowmap = _lim(rows, n_out)
        b.frag(fwd_hi, fwd_lo, offV[pl], n_in, ksf, ntf, 0, rowmap, kmap, 1.0, -1)
        if rowmap_r is not None:
            b.frag(rev_hi, rev_lo, offV[pl], n_in, ksr, ntr, 1, rowmap_r, _lim(slot_features(ksr), n_out), 1.0, -1)
        if with_bias:
            b.accvec(bias, offB[pl], 1, ntf, rowmap)
    jobs, maps, units = b.finish()
    return {"layout": ly, "jobs": jobs, "maps": maps, "units": units, "n_params": 0, "offW": offV, "offb": offB,
            "ins": NERF_IN, "outs": NERF_OUT, "n_raw": total_raw, "offB": offB, "offG": [None] * 12, "offV": offV,
            "rows": np.zeros(0, dtype=ROW_DTYPE), "bias_segs": np.zeros((0, 4), dtype=np.int32), "weight_norm": False,
            "names": NERF_NAMES}


# ---- Lvis (reference models/fields.py:338-369), plain nn.Linear layers, inference packs only -----------------------------
LVIS_IN = [90, 256, 256, 256, 256]
LVIS_OUT = [256, 256, 256, 256, 1]
LVIS_PTS_PE, LVIS_VIEW_PE = 63, 27


def build_lvis_jobs():
    """Pack tables of the stage-2 visibility network for fneus_lvis_visibility (csrc/lvis_kernels.hip, layout 3): forward packs
    and biases only.  Layer 0's k-slots: 64 slots for PE10(point) (reference columns 0..62), 32 slots for PE4(direction)
    (columns 63..89)."""
    ly = query_layout(3)
    offB, _offG, offV, total_raw = raw_offsets(LVIS_IN, LVIS_OUT, weight_norm=False)
    b = _Builder()
    for l in range(5):
        ksf, ntf, _ksr, _ntr = [int(v) for v in ly.geom[l]]
        fwd_hi, fwd_lo, _rh, _rl, bias = [int(v) for v in ly.off[l]]
        n_in, n_out = LVIS_IN[l], LVIS_OUT[l]
        rowmap = _lim(np.arange(ntf * 32), n_out)
        feat = slot_features(ksf)
        if l == 0:
            kmap = np.where(feat < LVIS_PTS_PE, feat, np.where((feat >= 64) & (feat - 64 < LVIS_VIEW_PE), LVIS_PTS_PE + feat - 64, -1))
        else:
            kmap = _lim(feat, n_in)
        b.frag(fwd_hi, fwd_lo, offV[l], n_in, ksf, ntf, 0, rowmap, kmap, 1.0, -1)
        b.accvec(bias, offB[l], 1, ntf, rowmap)
    # the single row of the last layer in accumulator layout (the two-pass kernel takes 256 -> 1 as a vector dot product)
    b.accvec(ly.extra, offV[4], 1, 8, np.arange(256))
    jobs, maps, units = b.finish()
    return {"layout": ly, "jobs": jobs, "maps": maps, "units": units, "n_params": 0, "offW": offV, "offb": offB,
            "ins": LVIS_IN, "outs": LVIS_OUT, "n_raw": total_raw, "offB": offB, "offG": [None] * 5, "offV": offV,
            "rows": np.zeros(0, dtype=ROW_DTYPE), "bias_segs": np.zeros((0, 4), dtype=np.int32), "weight_norm": False}
