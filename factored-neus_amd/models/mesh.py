"""Iso-surface extraction of the SDF (reference models/renderer.py:14-40 extract_fields / extract_geometry, which call
PyMCubes) and a PLY writer (the reference exports through trimesh, exp_runner.py:529-530).

PyMCubes / trimesh are not dependencies here.  The surface is extracted by MARCHING TETRAHEDRA on the same regular
grid, vectorised torch ops on the device that holds the grid: every cube is cut into the 6 tetrahedra around its main
diagonal (the cut is identical in every cube, so faces of neighbouring cubes agree and the mesh is watertight), a
tetrahedron yields 0, 1 or 2 triangles, vertices are placed by linear interpolation along grid edges exactly like
marching cubes does and welded by their (grid vertex, grid vertex) edge key.  The triangulation differs from PyMCubes'
(more, smaller triangles), the surface it samples is the same to O(h^2); SURVEY.md section 8(c) therefore pins the SDF
grid, not the vertex list.  Only cubes that straddle the threshold are ever expanded.
"""
from __future__ import annotations

import struct
from typing import Callable, Tuple

import numpy as np
import torch

# cube corner c -> offset (dx, dy, dz); tetrahedra share the diagonal corner 0 - corner 6
_CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
_TETS = [(0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6), (0, 5, 1, 6)]


def _case_table():
    """[16 inside-masks][2 triangles][3 corners][2 edge end points (local tetrahedron vertex)], and triangle counts"""
    tab = np.zeros((16, 2, 3, 2), dtype=np.int64)
    cnt = np.zeros(16, dtype=np.int64)
    for code in range(16):
        ins = [v for v in range(4) if code >> v & 1]
        out = [v for v in range(4) if not code >> v & 1]
        if len(ins) == 1:
            i = ins[0]
            tab[code, 0] = [(i, out[0]), (i, out[1]), (i, out[2])]
            cnt[code] = 1
        elif len(ins) == 3:
            o = out[0]
            tab[code, 0] = [(ins[0], o), (ins[1], o), (ins[2], o)]
            cnt[code] = 1
        elif len(ins) == 2:
            (i, j), (a, b) = ins, out
            tab[code, 0] = [(i, a), (i, b), (j, b)]
            tab[code, 1] = [(i, a), (j, b), (j, a)]
            cnt[code] = 2
    return tab, cnt


_TAB, _CNT = _case_table()


def marching_tetrahedra(u: torch.Tensor, threshold: float = 0.0, slab: int = 64) -> Tuple[torch.Tensor, torch.Tensor]:
    """u [X,Y,Z] scalar field (the reference passes u = -sdf) -> (vertices [V,3] float32 in GRID-INDEX coordinates,
    triangles [T,3] int64), on u's device.  "Inside" is u > threshold; triangles are wound so that their normals point
    to decreasing u (outward for u = -sdf), like PyMCubes."""
    assert u.dim() == 3
    dev = u.device
    X, Y, Z = u.shape
    u = u.float()
    inside = u > threshold
    # cubes whose 8 corners are not all on one side, found slab by slab (bounded temporaries at 512^3)
    active = []
    for x0 in range(0, X - 1, slab):
        x1 = min(x0 + slab, X - 1)
        s = inside[x0:x1 + 1]
        acc_and = torch.ones((x1 - x0, Y - 1, Z - 1), dtype=torch.bool, device=dev)
        acc_or = torch.zeros_like(acc_and)
        for dx, dy, dz in _CORNERS:
            c = s[dx:dx + x1 - x0, dy:dy + Y - 1, dz:dz + Z - 1]
            acc_and &= c
            acc_or |= c
        idx = torch.nonzero(acc_or & ~acc_and)
        idx[:, 0] += x0
        active.append(idx)
    cubes = torch.cat(active, 0)                                           # [K,3]
    if cubes.shape[0] == 0:
        return torch.zeros(0, 3, device=dev), torch.zeros(0, 3, dtype=torch.int64, device=dev)
    corners = torch.tensor(_CORNERS, dtype=torch.int64, device=dev)        # [8,3]
    cpos = cubes[:, None, :] + corners[None, :, :]                         # [K,8,3] grid coordinates of the corners
    cid = (cpos[..., 0] * Y + cpos[..., 1]) * Z + cpos[..., 2]             # [K,8] global grid-vertex ids
    cval = u.reshape(-1)[cid]                                              # [K,8]
    tets = torch.tensor(_TETS, dtype=torch.int64, device=dev)              # [6,4]
    tid = cid[:, tets].reshape(-1, 4)                                      # [K*6,4]
    tval = cval[:, tets].reshape(-1, 4)
    tin = tval > threshold
    code = (tin.long() * torch.tensor([1, 2, 4, 8], device=dev)).sum(-1)
    tab = torch.from_numpy(_TAB).to(dev)
    cnt = torch.from_numpy(_CNT).to(dev)
    tri_lo, tri_hi, tri_ref = [], [], []
    for k in range(2):                                                     # first / second triangle of a tetrahedron
        sel = torch.nonzero(cnt[code] > k).reshape(-1)
        if sel.numel() == 0:
            continue
        e = tab[code[sel], k]                                              # [S,3,2] local end points of the 3 cut edges
        ids, vals = tid[sel], tval[sel]
        a = torch.gather(ids, 1, e[..., 0])                                # [S,3] inside end (global id)
        b = torch.gather(ids, 1, e[..., 1])                                #        outside end
        tri_lo.append(a)
        tri_hi.append(b)
    a = torch.cat(tri_lo, 0)                                               # [T,3] inside end of each triangle corner's edge
    b = torch.cat(tri_hi, 0)
    nv = X * Y * Z
    key = torch.minimum(a, b) * nv + torch.maximum(a, b)                   # undirected grid edge
    uniq, inv = torch.unique(key.reshape(-1), return_inverse=True)
    lo, hi = uniq // nv, uniq % nv

    def coords(i):
        return torch.stack([i // (Y * Z), (i // Z) % Y, i % Z], -1).float()

    ulo, uhi = u.reshape(-1)[lo], u.reshape(-1)[hi]
    t = ((threshold - ulo) / (uhi - ulo)).clamp(0.0, 1.0)[:, None]
    verts = coords(lo) + t * (coords(hi) - coords(lo))
    tris = inv.reshape(-1, 3)
    # winding: normal must point from the inside end points towards the outside end points
    p = verts[tris]
    nrm = torch.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], dim=-1)
    out_dir = (coords(b) - coords(a)).sum(1)
    flip = (nrm * out_dir).sum(-1) < 0
    tris = torch.where(flip[:, None], tris[:, [0, 2, 1]], tris)
    # drop triangles that collapsed onto a grid vertex (field exactly at the threshold there)
    ok = (tris[:, 0] != tris[:, 1]) & (tris[:, 1] != tris[:, 2]) & (tris[:, 0] != tris[:, 2])
    return verts, tris[ok]


def extract_fields(bound_min, bound_max, resolution: int, query_func: Callable[[torch.Tensor], torch.Tensor],
                   device=None, as_numpy: bool = True):
    """renderer.py:14-29: the field on a regular grid, evaluated in 64^3 blocks.  query_func maps [M,3] points to [M]
    values.  Returns float32 [R,R,R] (numpy like the reference, or the device tensor)."""
    bound_min = torch.as_tensor(bound_min, dtype=torch.float32).reshape(3)
    bound_max = torch.as_tensor(bound_max, dtype=torch.float32).reshape(3)
    dev = torch.device(device) if device is not None else bound_min.device
    N = 64
    axes = [torch.linspace(float(bound_min[i]), float(bound_max[i]), resolution, device=dev) for i in range(3)]
    u = torch.zeros(resolution, resolution, resolution, dtype=torch.float32, device=dev)
    with torch.no_grad():
        for xi in range(0, resolution, N):
            for yi in range(0, resolution, N):
                for zi in range(0, resolution, N):
                    xs, ys, zs = axes[0][xi:xi + N], axes[1][yi:yi + N], axes[2][zi:zi + N]
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).contiguous()
                    val = query_func(pts).reshape(len(xs), len(ys), len(zs))
                    u[xi:xi + len(xs), yi:yi + len(ys), zi:zi + len(zs)] = val
    return u.cpu().numpy() if as_numpy else u


def extract_geometry(bound_min, bound_max, resolution: int, threshold: float, query_func, device=None):
    """renderer.py:32-40: (vertices [V,3] float in the coordinates of the bounding box, triangles [T,3] int) as numpy"""
    u = extract_fields(bound_min, bound_max, resolution, query_func, device=device, as_numpy=False)
    verts, tris = marching_tetrahedra(u, threshold)
    b_min = torch.as_tensor(bound_min, dtype=torch.float32).reshape(3).cpu().numpy()
    b_max = torch.as_tensor(bound_max, dtype=torch.float32).reshape(3).cpu().numpy()
    vertices = verts.cpu().numpy().astype(np.float64) / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :]
    return vertices, tris.cpu().numpy()


def write_ply(path: str, vertices: np.ndarray, triangles: np.ndarray):
    """binary little-endian PLY: float32 x y z per vertex, uchar 3 + int32 x 3 per face"""
    v = np.asarray(vertices, dtype="<f4").reshape(-1, 3)
    f = np.asarray(triangles, dtype="<i4").reshape(-1, 3)
    header = ("ply\nformat binary_little_endian 1.0\n"
              f"element vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
              f"element face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n")
    rec = np.empty(len(f), dtype=[("n", "u1"), ("i", "<i4", (3,))])
    rec["n"], rec["i"] = 3, f
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(v.tobytes())
        fh.write(rec.tobytes())


def read_ply(path: str):
    """reads what write_ply writes, plus binary / ascii point clouds and meshes with float or double x y z leading each
    vertex (enough for DTU's stl*_total.ply and for meshes exported by the reference) -> (vertices [V,3], faces [F,3] or None)"""
    with open(path, "rb") as fh:
        fmt, elems, cur = None, [], None
        while True:
            line = fh.readline().decode("ascii", "replace").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element"):
                cur = [line.split()[1], int(line.split()[2]), []]
                elems.append(cur)
            elif line.startswith("property"):
                cur[2].append(line.split()[1:])
            elif line == "end_header":
                break
        types = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
                 "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "short": "i2", "ushort": "u2"}
        verts, faces = None, None
        for name, count, props in elems:
            if fmt == "ascii":
                rows = [fh.readline().split() for _ in range(count)]
                if name == "vertex":
                    verts = np.array([[float(x) for x in r[:3]] for r in rows], dtype=np.float64)
                elif name == "face":
                    faces = np.array([[int(x) for x in r[1:4]] for r in rows], dtype=np.int64)
                continue
            end = "<" if fmt == "binary_little_endian" else ">"
            if name == "face" and props and props[0][0] == "list":
                dt = np.dtype([("n", end + types[props[0][1]]), ("i", end + types[props[0][2]], (3,))])
                raw = np.frombuffer(fh.read(dt.itemsize * count), dtype=dt)
                if count and not np.all(raw["n"] == 3):
                    raise ValueError("only triangle faces are supported")
                faces = raw["i"].astype(np.int64)
            else:
                dt = np.dtype([(p[-1] + str(k), end + types[p[0]]) for k, p in enumerate(props)])
                raw = np.frombuffer(fh.read(dt.itemsize * count), dtype=dt)
                if name == "vertex":
                    verts = np.stack([raw[dt.names[k]].astype(np.float64) for k in range(3)], -1)
    return verts, faces
