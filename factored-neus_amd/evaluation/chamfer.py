"""Chamfer-L1 between a reconstructed mesh and a reference point cloud: the metric half of BASELINE's headline number
(reference evaluation/dtu_eval.py:36-162, itself adapted from DTUeval-python).

Same procedure as the reference: (1) the mesh is turned into a point cloud by regular barycentric sampling of every
triangle at density `thresh` plus its vertices, (2) the cloud is thinned so that no two points are closer than `thresh`,
(3) optional DTU observability mask / ground plane, (4) nearest-neighbour distances in both directions, distances
>= max_dist dropped, overall = (mean data->reference + mean reference->data) / 2.
open3d / trimesh are not needed: PLY files go through models/mesh.py, neighbours through scipy's cKDTree.
Without DTU files (this image has none) the same code scores a mesh against an analytic or synthetic reference cloud:
`python -m evaluation.chamfer mesh.ply reference.ply`.
"""
from __future__ import annotations

import sys
from typing import Optional

import numpy as np
from scipy.spatial import cKDTree


def sample_mesh(vertices: np.ndarray, triangles: np.ndarray, thresh: float) -> np.ndarray:
    """dtu_eval.py:18-27, 55-77: vertices + a regular barycentric lattice on every triangle, spacing ~ thresh"""
    tri = vertices[triangles]
    v1, v2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    l1, l2 = np.linalg.norm(v1, axis=-1), np.linalg.norm(v2, axis=-1)
    area2 = np.linalg.norm(np.cross(v1, v2), axis=-1)
    ok = area2 > 0
    tri, v1, v2, l1, l2, area2 = tri[ok], v1[ok], v2[ok], l1[ok], l2[ok], area2[ok]
    thr = thresh * np.sqrt(l1 * l2 / area2)
    n1, n2 = np.floor(l1 / thr).astype(np.int64), np.floor(l2 / thr).astype(np.int64)
    pts = [vertices]
    # triangles with the same lattice size share one set of barycentric coordinates
    key = n1 * (n2.max() + 1 if len(n2) else 1) + n2
    for k in np.unique(key):
        sel = np.nonzero(key == k)[0]
        a, b = int(n1[sel[0]]), int(n2[sel[0]])
        c = np.mgrid[:a + 1, :b + 1].astype(np.float64) + 0.5
        c[0] /= max(a, 1e-7)
        c[1] /= max(b, 1e-7)
        c = c.transpose(1, 2, 0).reshape(-1, 2)
        c = c[c.sum(-1) < 1]
        if len(c) == 0:
            continue
        q = tri[sel, None, 0] + v1[sel, None] * c[None, :, :1] + v2[sel, None] * c[None, :, 1:]
        pts.append(q.reshape(-1, 3))
    return np.concatenate(pts, 0)


def thin(points: np.ndarray, thresh: float, seed: int = 0) -> np.ndarray:
    """dtu_eval.py:79-94: shuffle, then keep a point only if no kept point lies within thresh of it"""
    rng = np.random.default_rng(seed)
    pts = points[rng.permutation(len(points))]
    tree = cKDTree(pts)
    nbrs = tree.query_ball_point(pts, r=thresh)
    keep = np.ones(len(pts), dtype=bool)
    for i, idx in enumerate(nbrs):
        if keep[i]:
            keep[idx] = False
            keep[i] = True
    return pts[keep]


def chamfer_l1(data: np.ndarray, reference: np.ndarray, max_dist: float = 20.0, data_for_s2d: Optional[np.ndarray] = None,
               reference_for_s2d: Optional[np.ndarray] = None):
    """-> (mean data->reference, mean reference->data, overall); dtu_eval.py:124-141, 158.  The reference measures data ->
    reference against the WHOLE reference cloud (:121-126) and reference -> data from the part of it above the ground plane
    (:131-140): `reference_for_s2d`."""
    d2s, _ = cKDTree(reference).query(data, k=1)
    s2d, _ = cKDTree(data if data_for_s2d is None else data_for_s2d).query(reference if reference_for_s2d is None else reference_for_s2d, k=1)
    mean_d2s = float(d2s[d2s < max_dist].mean())
    mean_s2d = float(s2d[s2d < max_dist].mean())
    return mean_d2s, mean_s2d, 0.5 * (mean_d2s + mean_s2d)


def evaluate_mesh(vertices, triangles, reference_points, thresh: float = 0.2, max_dist: float = 20.0, obs_mask=None,
                  plane=None, patch: float = 60.0):
    """Full procedure.  obs_mask = (ObsMask bool grid, BB [2,3], Res) and plane = P [4] are the DTU extras
    (dtu_eval.py:100-117, 131-136); None skips them."""
    data = thin(sample_mesh(np.asarray(vertices, np.float64), np.asarray(triangles), thresh), thresh)
    data_in, data_obs = data, data
    if obs_mask is not None:
        mask, bb, res = obs_mask
        bb = np.asarray(bb, np.float32)
        inb = ((data >= bb[:1] - patch) & (data < bb[1:] + patch * 2)).sum(-1) == 3
        data_in = data[inb]
        grid = np.around((data_in - bb[:1]) / res).astype(np.int32)
        g_in = ((grid >= 0) & (grid < np.expand_dims(mask.shape, 0))).sum(-1) == 3
        gi = grid[g_in]
        data_obs = data_in[g_in][mask[gi[:, 0], gi[:, 1], gi[:, 2]].astype(bool)]
    ref = np.asarray(reference_points, np.float64)
    ref_above = None
    if plane is not None:
        hom = np.concatenate([ref, np.ones_like(ref[:, :1])], -1)
        ref_above = ref[(np.asarray(plane).reshape(1, 4) * hom).sum(-1) > 0]
    return chamfer_l1(data_obs, ref, max_dist, data_for_s2d=data_in, reference_for_s2d=ref_above)


def main(argv):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from models.mesh import read_ply
    if len(argv) < 3:
        print("usage: python -m evaluation.chamfer mesh.ply reference_points.ply [thresh] [max_dist]")
        return 2
    v, f = read_ply(argv[1])
    ref, _ = read_ply(argv[2])
    thresh = float(argv[3]) if len(argv) > 3 else 0.2
    max_dist = float(argv[4]) if len(argv) > 4 else 20.0
    d2s, s2d, overall = evaluate_mesh(v, f, ref, thresh, max_dist)
    print(f"{d2s} {s2d} {overall}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
