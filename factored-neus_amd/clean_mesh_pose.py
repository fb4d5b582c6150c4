#!/usr/bin/env python3
"""Mesh cleaning before the DTU evaluation, with the reference's command line (clean_mesh_pose.py:1-73).

    python clean_mesh_pose.py --scene 65 --setting wmask --suffix 00300000 [--data_root ./public_data/data_DTU] [--exp_root ./exp/data_DTU]

A vertex survives when its projection lies inside the (25 x 25 ellipse-dilated) object mask of EVERY camera
(clean_mesh_pose.py:22-46); faces with a removed vertex are dropped (:59-64) and the largest connected component is kept
(:68-69).  numpy / scipy only: the dilation is scipy.ndimage.binary_dilation with OpenCV's MORPH_ELLIPSE footprint, the
components come from scipy.sparse.csgraph, PLY files from models/mesh.py (cv2 / trimesh are not dependencies).
"""
from __future__ import annotations

import argparse
import os
import sys
from glob import glob

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def ellipse_footprint(k: int = 25) -> np.ndarray:
    """cv.getStructuringElement(cv.MORPH_ELLIPSE, (k, k)): row y spans |x - c| <= round(c * sqrt(1 - ((y - c) / c)^2))"""
    c = k // 2
    fp = np.zeros((k, k), dtype=bool)
    for y in range(k):
        dy = y - c
        dx = int(round(c * np.sqrt(max(0.0, 1.0 - (dy * dy) / float(c * c))))) if c > 0 else 0
        fp[y, max(c - dx, 0): min(c + dx, k - 1) + 1] = True
    return fp


def clean_points_by_mask(points: np.ndarray, world_mats, masks, dilate: int = 25) -> np.ndarray:
    """clean_mesh_pose.py:22-46.  points [V,3] world coordinates; world_mats: the cameras' 4 x 4 `world_mat_i`; masks: their
    [H,W] (or [H,W,3]) uint8 / bool object masks.  -> bool [V]"""
    from scipy.ndimage import binary_dilation
    inside = np.ones(len(points), dtype=bool)
    fp = ellipse_footprint(dilate)
    for P, m in zip(world_mats, masks):
        m = np.asarray(m)
        if m.ndim == 3:
            m = m[:, :, 0]
        m = m > 128 if m.dtype != bool else m
        H, W = m.shape
        m = binary_dilation(m, structure=fp)
        # one pixel of "inside" all around: points that project outside the image are not judged by this camera (:38-39)
        m = np.pad(m, 1, mode="constant", constant_values=True)
        pix = points @ P[:3, :3].T + P[:3, 3]
        pix = pix / pix[:, 2:]
        pix = np.round(pix).astype(np.int64) + 1
        inside &= m[pix[:, 1].clip(0, H + 1), pix[:, 0].clip(0, W + 1)]
    return inside


def largest_component(vertices: np.ndarray, faces: np.ndarray):
    """trimesh split(only_watertight=False) + the piece with most faces (clean_mesh_pose.py:68-69)"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    if len(faces) == 0:
        return vertices[:0], faces
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    g = coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(len(vertices), len(vertices)))
    _, label = connected_components(g, directed=False)
    face_label = label[faces[:, 0]]
    best = np.bincount(face_label).argmax()
    keep_f = face_label == best
    used = np.unique(faces[keep_f])
    remap = -np.ones(len(vertices), dtype=np.int64)
    remap[used] = np.arange(len(used))
    return vertices[used], remap[faces[keep_f]]


def clean_mesh(vertices: np.ndarray, faces: np.ndarray, world_mats, masks, dilate: int = 25):
    """clean_mesh_pose.py:49-71 -> (vertices, faces) of the cleaned mesh"""
    keep = clean_points_by_mask(vertices, world_mats, masks, dilate)
    index = -np.ones(len(vertices), dtype=np.int64)
    index[keep] = np.arange(int(keep.sum()))
    f_keep = keep[faces[:, 0]] & keep[faces[:, 1]] & keep[faces[:, 2]]
    return largest_component(vertices[keep], index[faces[f_keep]])


def main():
    from PIL import Image
    from models.mesh import read_ply, write_ply
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", type=str, required=True)
    ap.add_argument("--setting", type=str, required=True)
    ap.add_argument("--suffix", default="")
    ap.add_argument("--data_root", default="./public_data/data_DTU")
    ap.add_argument("--exp_root", default="./exp/data_DTU")
    args = ap.parse_args()
    scan, suffix = int(args.scene), int(args.suffix)
    case = os.path.join(args.data_root, "dtu_scan{}".format(scan))
    cams = np.load(os.path.join(case, "cameras_sphere.npz"))
    mask_lis = sorted(glob(os.path.join(case, "mask", "*.png")))
    n_images = len(mask_lis)
    world_mats = [cams["world_mat_{}".format(i)] for i in range(n_images)]
    masks = [np.asarray(Image.open(p).convert("RGB")) for p in mask_lis]
    old_dir = os.path.join(args.exp_root, "dtu_scan{}".format(scan), args.setting, "meshes")
    new_dir = os.path.join(args.exp_root, "dtu_scan{}".format(scan), args.setting, "meshes_clean")
    os.makedirs(new_dir, exist_ok=True)
    v, f = read_ply(os.path.join(old_dir, "{:0>8d}.ply".format(suffix)))
    v2, f2 = clean_mesh(np.asarray(v, dtype=np.float64), np.asarray(f, dtype=np.int64), world_mats, masks)
    out = os.path.join(new_dir, "{:0>8d}.ply".format(suffix))
    write_ply(out, v2, f2)
    print("{}: {} -> {} vertices, {} -> {} faces".format(out, len(v), len(v2), len(f), len(f2)))


if __name__ == "__main__":
    main()
