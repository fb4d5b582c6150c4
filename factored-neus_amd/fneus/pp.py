"""Fragment planes ("PP planes", csrc/fneus_pp.h): host-side index maps of the layout, used to build inputs for and to
read back the outputs of the kernels in tests and tools.  No maths here; the product path never packs on the host.

plane[tile][F fragments][64 slots][8 bf16]: element (sample n, feature f) lives at
    tile = n >> 5, r = n & 31, ks = f >> 4, h = (f >> 2) & 1, j = 4 * ((f >> 3) & 1) + (f & 3),
    slot = (2 r + h) ^ (8 (ks & 1)).
"""
from __future__ import annotations

import functools

import numpy as np
import torch


@functools.lru_cache(maxsize=None)
def _index(F: int):
    """flat index into a [F, 64, 8] block for every (r, f) of a [32, 16 F] tile"""
    r = np.arange(32)[:, None]
    f = np.arange(16 * F)[None, :]
    ks, h, j = f >> 4, (f >> 2) & 1, 4 * ((f >> 3) & 1) + (f & 3)
    slot = (2 * r + h) ^ (8 * (ks & 1))
    return torch.from_numpy(((ks * 64 + slot) * 8 + j).astype(np.int64))       # [32, 16 F]


def n_tiles(n: int) -> int:
    """sample tiles that hold data"""
    return (n + 31) // 32


def alloc_tiles(n: int) -> int:
    """sample tiles a plane is allocated for (csrc/fneus_pp.h pp_tiles): an even count, because workgroups that carry two
    tiles store a whole padding tile of zeros behind a ragged end"""
    return 2 * ((n + 63) // 64)


def pack(x: torch.Tensor, F: int, planes: int = 1):
    """x [N, W <= 16 F] float -> bf16 planes [planes, tiles, F, 64, 8] (hi, then lo = bf16(x - hi)); padding is zero"""
    N, W = x.shape
    T = alloc_tiles(N)
    full = torch.zeros(T * 32, 16 * F, dtype=torch.float32, device=x.device)
    full[:N, :W] = x.float()
    idx = _index(F).to(x.device)
    out = torch.zeros(planes, T, F * 64 * 8, dtype=torch.bfloat16, device=x.device)
    hi = full.bfloat16()
    parts = [hi] + ([(full - hi.float()).bfloat16()] if planes == 2 else [])
    for p, v in enumerate(parts):
        out[p].scatter_(1, idx.reshape(1, -1).expand(T, -1), v.reshape(T, -1))
    return out.reshape(planes, T, F, 64, 8)


def unpack(plane: torch.Tensor, n: int | None = None) -> torch.Tensor:
    """one plane [tiles, F, 64, 8] (bf16 or int16 ...) -> [tiles * 32 (or n), 16 F] of the same dtype"""
    T, F = plane.shape[0], plane.shape[1]
    idx = _index(F).to(plane.device)
    flat = plane.reshape(T, F * 64 * 8)
    out = flat.gather(1, idx.reshape(1, -1).expand(T, -1)).reshape(T * 32, 16 * F)
    return out if n is None else out[:n]


def value(planes: torch.Tensor, n: int | None = None) -> torch.Tensor:
    """fp32 value of [planes, tiles, F, 64, 8] bf16 planes (hi + lo)"""
    return sum(unpack(planes[p], n).float() for p in range(planes.shape[0]))
