"""Iso-surface extraction (models/mesh.py, the build's stand-in for PyMCubes: reference renderer.py:14-40) and the
Chamfer harness (evaluation/chamfer.py, reference evaluation/dtu_eval.py) on analytic SDFs -- SURVEY.md section 8(c):
mesh vertices are unpinned by the reference, so the extractor is validated on surfaces whose answer is known."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))


def _sphere_mesh(res=40, radius=0.7, centre=(0.05, -0.1, 0.02)):
    from models.mesh import extract_geometry
    c = torch.tensor(centre)
    q = lambda pts: -(torch.linalg.norm(pts - c, dim=-1) - radius)          # the reference queries u = -sdf
    return extract_geometry(torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0]), res, 0.0, q), np.array(centre), radius


def test_sphere_vertices_lie_on_the_surface_and_mesh_is_closed():
    (v, f), c, r = _sphere_mesh()
    h = 2.0 / 39
    d = np.linalg.norm(v - c, axis=1) - r
    assert np.abs(d).max() < 0.5 * h * h / r + 1e-6            # linear interpolation of a curved field: O(h^2 / r)
    # watertight 2-manifold: every undirected edge belongs to exactly two triangles, consistently oriented
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    und = np.sort(e, 1)
    _, counts = np.unique(und, axis=0, return_counts=True)
    assert (counts == 2).all()
    _, dcounts = np.unique(e, axis=0, return_counts=True)
    assert (dcounts == 1).all()                                 # each directed edge once -> consistent winding
    assert len(v) - len(und) // 2 + len(f) == 2                 # Euler characteristic of a sphere
    # outward normals (u = -sdf decreases outwards) and area -> 4 pi r^2
    p = v[f]
    n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    assert (np.einsum("ij,ij->i", n, p.mean(1) - c) > 0).all()
    area = 0.5 * np.linalg.norm(n, axis=1).sum()
    assert abs(area - 4 * np.pi * r * r) / (4 * np.pi * r * r) < 5e-3


def test_threshold_and_box_mapping():
    from models.mesh import extract_geometry
    q = lambda pts: -(pts.abs().max(dim=-1)[0] - 0.5)                         # a cube of half-size 0.5, u = -sdf
    v, f = extract_geometry([-1.0, -0.8, -0.9], [1.0, 0.9, 1.1], 33, -0.1, q)   # level u = -0.1 -> half-size 0.6
    assert len(f) > 0
    assert np.abs(np.abs(v).max(1) - 0.6).max() < 0.08
    assert np.abs(v).max() <= 0.6 + 1e-5


def test_empty_field_gives_empty_mesh():
    from models.mesh import marching_tetrahedra
    v, f = marching_tetrahedra(torch.full((8, 8, 8), -1.0), 0.0)
    assert v.shape == (0, 3) and f.shape == (0, 3)


def test_ply_round_trip(tmp_path):
    from models.mesh import write_ply, read_ply
    (v, f), _, _ = _sphere_mesh(res=16)
    path = str(tmp_path / "m.ply")
    write_ply(path, v, f)
    v2, f2 = read_ply(path)
    assert np.allclose(v2, v.astype(np.float32)) and np.array_equal(f2, f)


def test_chamfer_of_extracted_sphere_against_analytic_points():
    from evaluation.chamfer import evaluate_mesh, sample_mesh, thin, chamfer_l1
    (v, f), c, r = _sphere_mesh(res=48)
    rs = np.random.RandomState(0)
    g = rs.standard_normal((60000, 3))
    ref = c + r * g / np.linalg.norm(g, axis=1, keepdims=True)
    d2s, s2d, overall = evaluate_mesh(v, f, ref, thresh=0.01, max_dist=1.0)
    assert d2s < 8e-3 and s2d < 8e-3          # = the spacing of the two point sets, not a surface error
    assert abs(overall - 0.5 * (d2s + s2d)) < 1e-12
    # a sphere that is 0.05 too large is 0.05 away in both directions
    ref_big = c + (r + 0.05) * g / np.linalg.norm(g, axis=1, keepdims=True)
    d2s_b, s2d_b, _ = evaluate_mesh(v, f, ref_big, thresh=0.01, max_dist=1.0)
    assert abs(d2s_b - 0.05) < 5e-3 and abs(s2d_b - 0.05) < 5e-3
    # thinning leaves no pair closer than the threshold; max_dist drops outliers
    pts = thin(sample_mesh(v, f, 0.05), 0.05)
    from scipy.spatial import cKDTree
    dd, _ = cKDTree(pts).query(pts, k=2)
    assert dd[:, 1].min() >= 0.05
    far = np.concatenate([pts, [[50.0, 0, 0]]], 0)
    a, _, _ = chamfer_l1(far, ref, max_dist=1.0)
    b, _, _ = chamfer_l1(pts, ref, max_dist=1.0)
    assert abs(a - b) < 1e-12


def test_clean_mesh_by_masks_and_largest_component():
    """clean_mesh_pose.py:22-71: vertices outside any camera's dilated mask go, faces with them, then small components"""
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
    import clean_mesh_pose as C
    fp = C.ellipse_footprint(25)
    assert fp.shape == (25, 25) and fp[12].all() and fp[0].sum() == 1 and fp[:, 12].all() and fp.sum() == fp[::-1].sum()
    # two cameras looking down the z and x axes at a unit cube of vertices; masks = discs of radius 20 px around the centre
    K = np.array([[100.0, 0, 50], [0, 100.0, 50], [0, 0, 1]])

    def world_mat(R, t):
        P = np.eye(4)
        P[:3, :4] = K @ np.concatenate([R, t[:, None]], 1)
        return P

    Rz, Rx = np.eye(3), np.array([[0.0, 0, -1], [0, 1, 0], [1, 0, 0]])
    mats = [world_mat(Rz, np.array([0.0, 0, 5])), world_mat(Rx, np.array([0.0, 0, 5]))]
    yy, xx = np.mgrid[0:100, 0:100]
    disc = (((xx - 50) ** 2 + (yy - 50) ** 2) <= 20 ** 2).astype(np.uint8) * 255
    masks = [disc, disc]
    # a strip of triangles along x from -3 to 3 (projects from 50-60 to 50+60 px in camera 0: the ends fall outside the
    # dilated disc of radius 20 + 12), plus one far-away stray triangle inside both masks' padding... and one small island
    xs = np.linspace(-3, 3, 61)
    v = np.stack([np.stack([xs, -0.05 * np.ones_like(xs), np.zeros_like(xs)], 1),
                  np.stack([xs, 0.05 * np.ones_like(xs), np.zeros_like(xs)], 1)], 1).reshape(-1, 3)
    f = []
    for i in range(60):
        a, b, c, d = 2 * i, 2 * i + 1, 2 * i + 2, 2 * i + 3
        f += [[a, b, c], [b, d, c]]
    island = np.array([[0.0, 1.0, 0.0], [0.1, 1.0, 0.0], [0.0, 1.1, 0.0]])      # inside both masks, not connected to the strip
    v = np.concatenate([v, island], 0)
    f = np.array(f + [[len(v) - 3, len(v) - 2, len(v) - 1]])
    keep = C.clean_points_by_mask(v, mats, masks)
    # camera 0 sees x as u = 50 + 100 x / 5: inside the dilated disc (radius 32) for |x| <= 1.6; vertices that project
    # outside the image (u < -0.5 or u > 99.5) are not judged by that camera (the reference pads the mask with ones)
    expect = (np.abs(xs) <= 1.6 + 1e-9) | (xs <= -2.6 + 1e-9) | (xs >= 2.5 - 1e-9)
    assert keep[: len(xs) * 2].reshape(-1, 2)[:, 0].tolist() == expect.tolist()
    assert keep[-3:].all()
    v2, f2 = C.clean_mesh(v, f, mats, masks)
    assert np.abs(v2[:, 0]).max() <= 1.6 + 1e-9 and np.abs(v2[:, 1]).max() < 0.5        # the strip's middle, without the island
    assert len(f2) == 2 * (int((np.abs(xs) <= 1.6 + 1e-9).sum()) - 1) and f2.max() == len(v2) - 1


def test_chamfer_pinned_to_the_reference_evaluation(golden_dir):
    """evaluation/chamfer.py against the numbers of the reference's own evaluation/dtu_eval.py eval() on the synthetic
    DTU-shaped case of fneus.synth.dtu_eval_scene (mesh sampling, thinning, observability mask, ground plane, both directed
    means; tests/golden/gen_golden.py gen_dtu_eval).  The thinning depends on a random shuffle (unseeded in the reference):
    two shuffles move the means by a few 1e-4 here, the bound is 1 %."""
    import os
    from evaluation.chamfer import evaluate_mesh
    from fneus import synth
    g = np.load(os.path.join(golden_dir, "dtu_eval_synth.npz"))
    sc = synth.dtu_eval_scene(int(g["scene_seed"]))
    d2s, s2d, overall = evaluate_mesh(sc["vertices"], sc["triangles"], sc["stl"], thresh=0.2, max_dist=20.0,
                                      obs_mask=(sc["ObsMask"], sc["BB"], sc["Res"]), plane=sc["P"], patch=60.0)
    print(f"  Chamfer d2s {d2s:.5f} (reference {float(g['mean_d2s']):.5f})  s2d {s2d:.5f} ({float(g['mean_s2d']):.5f})")
    assert abs(d2s - float(g["mean_d2s"])) <= 0.01 * float(g["mean_d2s"])
    assert abs(s2d - float(g["mean_s2d"])) <= 0.01 * float(g["mean_s2d"])
    assert abs(overall - float(g["over_all"])) <= 0.01 * float(g["over_all"])
