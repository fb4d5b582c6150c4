"""Thin torch-tensor wrappers over the C ABI (device memory + stream plumbing only; no maths here)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, netdesc
from ._lib import lib, check

PREC_FAST, PREC_PARITY = 1, 3


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_f32(t: torch.Tensor, name: str):
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 CUDA tensor")


class PackedNet:
    """Device-side packed weights of one MLP ('sdf' or 'color') + its pack-job tables."""

    def __init__(self, kind: str, device):
        desc = netdesc.build_sdf_jobs() if kind == "sdf" else netdesc.build_color_jobs()
        self.kind, self.desc, self.device = kind, desc, device
        self.layout = desc["layout"]
        self.n_params = desc["n_params"]
        self.jobs = torch.from_numpy(desc["jobs"].view(np.uint8).copy()).to(device)
        self.maps = torch.from_numpy(desc["maps"]).to(device)
        self.n_jobs = len(desc["jobs"])
        self.units = desc["units"]
        self.blob = torch.zeros(self.layout.total, dtype=torch.uint8, device=device)

    def flat_from_lists(self, Ws, bs) -> torch.Tensor:
        parts = []
        for W, b in zip(Ws, bs):
            parts.append(W.reshape(-1))
            parts.append(b.reshape(-1))
        flat = torch.cat(parts)
        assert flat.numel() == self.n_params
        return flat

    def split_flat(self, flat: torch.Tensor):
        Ws, bs = [], []
        for l, (i, o) in enumerate(zip(self.desc["ins"], self.desc["outs"])):
            Ws.append(flat[self.desc["offW"][l]: self.desc["offW"][l] + i * o].view(o, i))
            bs.append(flat[self.desc["offb"][l]: self.desc["offb"][l] + o])
        return Ws, bs

    def pack(self, flat: torch.Tensor):
        _chk_f32(flat, "flat params")
        assert flat.numel() == self.n_params
        check(lib.fneus_pack(_ptr(self.jobs), self.n_jobs, self.units, _ptr(self.maps), _ptr(flat), _ptr(self.blob),
                             _stream()), "fneus_pack")
        return self.blob


class SdfStash:
    """bf16 activation planes written by sdf_fwd_grad (see include/fneus.h FneusSdfStash)."""

    def __init__(self, n: int, device, prec: int, train: bool):
        self.n, self.prec = n, prec
        bf = torch.bfloat16
        planes = 2 if prec == 3 else 1

        def alloc(*shape):
            return torch.empty((planes,) + shape, dtype=bf, device=device)

        self.pe = alloc(n, 48)
        self.h = alloc(8, n, 256)
        self.a = alloc(8, n, 256) if train else None
        self.feat = alloc(n, 256) if train else None
        s = _lib.FneusSdfStash()
        for name, t in (("pe", self.pe), ("h", self.h), ("a", self.a), ("feat", self.feat)):
            if t is None:
                continue
            setattr(s, name + "_hi", t[0].data_ptr())
            setattr(s, name + "_lo", t[1].data_ptr() if planes == 2 else None)
        self.c = s

    def plane(self, t):
        """fp32 value of a stash tensor (hi + lo)"""
        return t.float().sum(0)


def sdf_fwd(blob, n_pts: int, prec: int, pts=None, rays_o=None, rays_d=None, t=None, m: int = 1, out=None):
    dev = blob.device
    if out is None:
        out = torch.empty(n_pts, dtype=torch.float32, device=dev)
    check(lib.fneus_sdf_fwd(_ptr(blob), _ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(t), m, n_pts, _ptr(out), prec,
                            _stream()), "fneus_sdf_fwd")
    return out


def sdf_fwd_grad(blob, n_pts: int, prec: int, stash: SdfStash, train: bool, pts=None, rays_o=None, rays_d=None,
                 t=None, m: int = 1):
    dev = blob.device
    sdf = torch.empty(n_pts, dtype=torch.float32, device=dev)
    feat = torch.empty(n_pts, 256, dtype=torch.float32, device=dev)
    normal = torch.empty(n_pts, 3, dtype=torch.float32, device=dev)
    check(lib.fneus_sdf_fwd_grad(_ptr(blob), _ptr(pts), _ptr(rays_o), _ptr(rays_d), _ptr(t), m, n_pts,
                                 C.byref(stash.c), _ptr(sdf), _ptr(feat), _ptr(normal), prec, int(train), _stream()),
          "fneus_sdf_fwd_grad")
    return sdf, feat, normal
