"""Ray feeders for the stage-1 hot path.

Dataset           : DTU-format scenes (reference models/dataset.py:41-196): image/*.png, mask/*.png,
                    cameras_sphere.npz (world_mat_i, scale_mat_i).  Images and cameras live ON THE DEVICE and rays are
                    generated there, so a training step has no host->device copy (the reference indexes CPU images and
                    uploads every step, dataset.py:133-151).  Colours are BGR/256 like the reference (cv2 order).
SyntheticDataset  : DTU-shaped synthetic scene (two analytic spheres traced per pixel, cameras on a sphere) for smoke runs
                    and the end-to-end reconstruction test.
DatasetShiny      : Blender-format scenes of the womask configuration (reference models/dataset.py:522-662, Shiny Blender):
                    transforms_{split}.json, <frame>.png (sRGB -> linear by the 2.2 power), masks from <frame>_disp.tiff
                    (or <frame>_alpha.png for the "ball" scene), camera centres halved, OpenGL -> OpenCV axes.
The other reference loaders (Sk3d / Glossy*) are data-format variety outside the hot path and are not provided.
"""
from __future__ import annotations

import os
from glob import glob

import numpy as np
import torch


def load_K_Rt_from_P(P):
    """K, pose from a 3x4 projection (reference dataset.py:17-38 uses cv2.decomposeProjectionMatrix; here RQ)."""
    from scipy.linalg import rq
    M = P[:3, :3]
    K, R = rq(M)
    S = np.diag(np.sign(np.diag(K)))
    K, R = K @ S, S @ R
    if np.linalg.det(R) < 0:
        R = -R
    c = -np.linalg.inv(M) @ P[:3, 3]
    K = K / K[2, 2]
    intrinsics = np.eye(4, dtype=np.float32)
    intrinsics[:3, :3] = K
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = c
    return intrinsics, pose


class _RayMixin:
    """Ray generation on the HIP kernels (fneus_gen_rays_grid / fneus_gen_random_rays): images, masks and cameras are device
    resident, a training batch is one launch.  There is no host implementation: the loaders need a GPU device."""

    def _need_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("ray generation runs on the HIP kernels: construct the dataset on a cuda device")

    def gen_rays_at(self, img_idx, resolution_level=1):
        """all rays of one camera, [H/l, W/l, 3] each (dataset.py:115-131)"""
        from fneus import ops
        self._need_gpu()
        l = resolution_level
        tx = torch.linspace(0, self.W - 1, self.W // l, device=self.device)
        ty = torch.linspace(0, self.H - 1, self.H // l, device=self.device)
        return ops.gen_rays_grid(self.intrinsics_all_inv[int(img_idx)].contiguous(), self.pose_all[int(img_idx)].contiguous(), tx, ty)

    def gen_random_rays_at(self, img_idx, batch_size, pixels=None):
        """[B,10] = rays_o, rays_d, rgb, mask of random pixels of one camera (dataset.py:133-151), all on the device.
        pixels = (px, py) int64 tensors: use these instead of drawing (tests)"""
        from fneus import ops
        self._need_gpu()
        img_idx = int(img_idx)
        if pixels is None:
            px = torch.randint(low=0, high=self.W, size=[batch_size], device=self.device)
            py = torch.randint(low=0, high=self.H, size=[batch_size], device=self.device)
        else:
            px, py = pixels
        return ops.gen_random_rays(self.intrinsics_all_inv[img_idx].contiguous(), self.pose_all[img_idx].contiguous(),
                                   self.images[img_idx], self.masks[img_idx], px, py)

    @staticmethod
    def near_far_from_sphere(rays_o, rays_d):
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return mid - 1.0, mid + 1.0


class Dataset(_RayMixin):
    def __init__(self, conf, device=None):
        from PIL import Image
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.conf = conf
        self.data_dir = conf.get_string("data_dir")
        cams = np.load(os.path.join(self.data_dir, conf.get_string("render_cameras_name")))
        self.images_lis = sorted(glob(os.path.join(self.data_dir, "image/*.png")))
        self.masks_lis = sorted(glob(os.path.join(self.data_dir, "mask/*.png")))
        self.n_images = len(self.images_lis)

        def read(path):     # RGB file -> BGR array / 256, like cv2.imread (dataset.py:61-63)
            return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)[..., ::-1] / 256.0

        self.images = torch.from_numpy(np.stack([read(p) for p in self.images_lis]).copy()).to(self.device)
        self.masks = torch.from_numpy(np.stack([read(p) for p in self.masks_lis]).copy()).to(self.device)
        self.scale_mats_np = [cams["scale_mat_%d" % i].astype(np.float32) for i in range(self.n_images)]
        self.world_mats_np = [cams["world_mat_%d" % i].astype(np.float32) for i in range(self.n_images)]
        intr, poses = [], []
        for s, w in zip(self.scale_mats_np, self.world_mats_np):
            K, pose = load_K_Rt_from_P((w @ s)[:3, :4])
            intr.append(torch.from_numpy(K))
            poses.append(torch.from_numpy(pose))
        self.intrinsics_all = torch.stack(intr).to(self.device)
        self.intrinsics_all_inv = torch.inverse(self.intrinsics_all)
        self.pose_all = torch.stack(poses).to(self.device)
        self.focal = self.intrinsics_all[0][0, 0]
        self.H, self.W = self.images.shape[1], self.images.shape[2]
        self.image_pixels = self.H * self.W
        self.object_bbox_min = np.array([-1.01, -1.01, -1.01], dtype=np.float32)
        self.object_bbox_max = np.array([1.01, 1.01, 1.01], dtype=np.float32)

    def image_at(self, idx, resolution_level):
        img = (self.images[idx] * 256.0).clip(0, 255)
        l = resolution_level
        return img[::l, ::l].cpu().numpy()


def _resize_bilinear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv.resize(img, (out_w, out_h)) with INTER_LINEAR on a float image [H, W, C] (no antialiasing, border replicated)"""
    h, w = img.shape[:2]
    if (out_h, out_w) == (h, w):
        return img.copy()

    def axis(n_in, n_out):
        x = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(x).astype(np.int64)
        f = (x - i0).astype(np.float32)
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f

    y0, y1, fy = axis(h, out_h)
    x0, x1, fx = axis(w, out_w)
    fx = fx[None, :, None]
    top = img[y0][:, x0] * (1.0 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1.0 - fx) + img[y1][:, x1] * fx
    fy = fy[:, None, None]
    return (top * (1.0 - fy) + bot * fy).astype(img.dtype)


class DatasetShiny(_RayMixin):
    """reference models/dataset.py:522-662.  File formats through PIL (imageio / tifffile / cv2 are not dependencies): PNG
    colours / 255 then ** 2.2 (rend_util.py:10-17), disparity TIFFs thresholded at 1e-6, `_alpha.png` masks / 256 thresholded
    at 0.5 and averaged over the channels.  Images, masks and cameras end up on the device; rays come from the HIP kernels."""

    def __init__(self, conf, frame_skip=1, split="train", device=None):
        import json
        from PIL import Image
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.conf, self.split = conf, split
        self.data_dir = conf.get_string("data_dir")
        with open(os.path.join(self.data_dir, "transforms_{}.json".format(split)), "r") as fp:
            meta = json.load(fp)
        ball = "ball" in self.data_dir
        image_paths, mask_paths, poses = [], [], []
        for frame in meta["frames"]:
            poses.append(np.array(frame["transform_matrix"], dtype=np.float64))
            image_paths.append(os.path.join(self.data_dir, frame["file_path"] + ".png"))
            mask_paths.append(os.path.join(self.data_dir, frame["file_path"] + ("_alpha.png" if ball else "_disp.tiff")))

        def load_rgb(path):                                   # rend_util.py:10-17
            img = np.asarray(Image.open(path), dtype=np.float32)[:, :, :3]
            return np.power(img / 255.0, 2.2).astype(np.float32)

        def load_mask(path):                                  # dataset.py:581-589
            if ball:
                m = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 256.0
                m[m > 0.5] = 1.0
                return np.mean(m, axis=-1).astype(np.float32)
            m = np.array(Image.open(path), dtype=np.float32)
            if m.ndim == 3:
                m = m[..., 0]
            m[m > 1e-6] = 1.0
            return m

        img_h, img_w = load_rgb(image_paths[0]).shape[:2]
        focal = 0.5 * img_w / np.tan(0.5 * float(meta["camera_angle_x"]))
        poses = np.array(poses)
        poses[..., 3] /= 2.0                                  # dataset.py:556-557: the whole last column, as the reference does
        # frame_skip thins images, poses AND masks (the reference thins the first two only and then fails in the reshape of its
        # mask stack, dataset.py:563-565, 603: with frame_skip > 1 image i must meet mask i * frame_skip)
        image_paths, poses, mask_paths = image_paths[::frame_skip], poses[::frame_skip, ...], mask_paths[::frame_skip]
        self.image_paths = self.images_lis = image_paths
        K = np.array([[focal, 0, img_w / 2], [0, focal, img_h / 2], [0, 0, 1]], dtype=np.float32)
        self.n_images = len(image_paths)
        self.images = torch.from_numpy(np.stack([load_rgb(p) for p in image_paths])).to(self.device)
        masks = np.stack([load_mask(p) for p in mask_paths])
        self.masks = torch.from_numpy(masks).reshape(self.n_images, img_h, img_w, 1).repeat(1, 1, 1, 3).to(self.device)
        # 4 x 4 intrinsics (the kernels take the DTU loader's layout); the reference keeps 3 x 3
        K4 = np.eye(4, dtype=np.float32)
        K4[:3, :3] = K
        self.intrinsics_all = torch.from_numpy(np.stack([K4] * self.n_images)).to(self.device)
        self.intrinsics_all_inv = torch.inverse(self.intrinsics_all)
        self.focal = focal
        convert = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0]))     # OpenGL camera axes -> OpenCV (dataset.py:592-597, 607)
        self.pose_all = (torch.from_numpy(poses).float() @ convert).contiguous().to(self.device)
        self.H, self.W = img_h, img_w
        self.images_pixels = self.image_pixels = self.H * self.W
        self.scale_mats_np = [np.eye(4, dtype=np.float32)] * self.n_images
        self.object_bbox_min = np.array([-1.01, -1.01, -1.01], dtype=np.float32)
        self.object_bbox_max = np.array([1.01, 1.01, 1.01], dtype=np.float32)

    def image_at(self, idx, resolution_level):
        """dataset.py:660-662: back to display gamma, resized to (H // l, W // l) like cv.resize's default (bilinear, pixel
        centres aligned: source coordinate (i + 0.5) * scale - 0.5, clamped at the border)"""
        img = np.power(self.images[idx].cpu().numpy(), 1.0 / 2.2) * 255
        return _resize_bilinear(img, self.H // resolution_level, self.W // resolution_level).clip(0, 255)


def export_shiny_scene(ds: "SyntheticDataset", out_dir: str, split: str = "train", ball: bool = False) -> str:
    """Write a SyntheticDataset in the Shiny-Blender layout DatasetShiny reads: transforms_{split}.json with OpenGL camera
    matrices whose centres are twice the unit-sphere ones, r_%d.png (sRGB-encoded so that the loader's 2.2 power returns
    ds.images) and the masks as r_%d_disp.tiff (float disparity) or r_%d_alpha.png."""
    import json
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    K = ds.intrinsics_all[0].cpu().numpy()
    frames = []
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    for i in range(ds.n_images):
        img = ds.images[i].cpu().numpy().astype(np.float64)
        png = np.clip(np.round(np.power(np.clip(img, 0.0, 1.0), 1.0 / 2.2) * 255.0), 0, 255).astype(np.uint8)
        Image.fromarray(png).save(os.path.join(out_dir, "r_%d.png" % i))
        m = ds.masks[i].cpu().numpy()[..., 0].astype(np.float32)
        if ball:
            Image.fromarray((m * 255).astype(np.uint8)).convert("RGB").save(os.path.join(out_dir, "r_%d_alpha.png" % i))
        else:
            Image.fromarray(m * 0.37, mode="F").save(os.path.join(out_dir, "r_%d_disp.tiff" % i))
        pose = ds.pose_all[i].cpu().numpy().astype(np.float64) @ flip       # OpenCV -> OpenGL axes
        pose[:3, 3] *= 2.0
        pose[3, 3] = 2.0            # the loader halves the whole last column (dataset.py:556-557): keep w = 1 after it
        frames.append({"file_path": "r_%d" % i, "transform_matrix": pose.tolist()})
    meta = {"camera_angle_x": float(2.0 * np.arctan(0.5 * ds.W / K[0, 0])), "frames": frames}
    with open(os.path.join(out_dir, "transforms_%s.json" % split), "w") as fp:
        json.dump(meta, fp)
    return out_dir


# analytic test scene inside the unit sphere: the union of two diffuse spheres (centre, radius, albedo in the image's
# channel order) under one directional light -- something a few hundred training steps can actually fit
SCENE_SPHERES = [((0.15, 0.0, 0.0), 0.45, (0.8, 0.3, 0.2)), ((-0.30, 0.20, 0.10), 0.30, (0.2, 0.5, 0.8))]
SCENE_LIGHT = (0.3, 0.5, 1.0)


def scene_sdf(p: np.ndarray) -> np.ndarray:
    """signed distance of the analytic scene (exact outside, a lower bound inside the overlap)"""
    return np.min([np.linalg.norm(p - np.asarray(c), axis=-1) - r for c, r, _ in SCENE_SPHERES], axis=0)


def scene_surface_points(n: int, seed: int = 0) -> np.ndarray:
    """~n points uniformly on the visible surface of the union (for Chamfer against extracted meshes)"""
    rs = np.random.RandomState(seed)
    area = np.array([r * r for _, r, _ in SCENE_SPHERES])
    out = []
    for k, (c, r, _) in enumerate(SCENE_SPHERES):
        g = rs.standard_normal((int(2 * n * area[k] / area.sum()), 3))
        p = np.asarray(c) + r * g / np.linalg.norm(g, axis=1, keepdims=True)
        keep = np.ones(len(p), dtype=bool)
        for j, (c2, r2, _) in enumerate(SCENE_SPHERES):
            if j != k:
                keep &= np.linalg.norm(p - np.asarray(c2), axis=1) > r2
        out.append(p[keep])
    p = np.concatenate(out, 0)
    return p[rs.permutation(len(p))[:n]]


def render_scene(rays_o: np.ndarray, rays_d: np.ndarray):
    """closest ray / sphere hit -> (rgb [M,3], mask [M]); rays_d unit length"""
    best_t = np.full(len(rays_o), np.inf)
    rgb = np.zeros((len(rays_o), 3), dtype=np.float64)
    light = np.asarray(SCENE_LIGHT) / np.linalg.norm(SCENE_LIGHT)
    for c, r, albedo in SCENE_SPHERES:
        oc = rays_o - np.asarray(c)
        b = (oc * rays_d).sum(-1)
        disc = b * b - ((oc * oc).sum(-1) - r * r)
        t = -b - np.sqrt(np.maximum(disc, 0.0))
        hit = (disc > 0) & (t > 0) & (t < best_t)
        nrm = (oc + rays_d * t[:, None]) / r
        shade = 0.35 + 0.65 * np.maximum((nrm * light).sum(-1), 0.0)
        rgb[hit] = np.asarray(albedo)[None, :] * shade[hit, None]
        best_t[hit] = t[hit]
    return rgb.astype(np.float32), np.isfinite(best_t).astype(np.float32)


class SyntheticDataset(_RayMixin):
    """DTU-shaped synthetic scene: n_images pinhole cameras on a sphere of radius 2.8 looking at the origin, images and
    masks rendered analytically from SCENE_SPHERES (no files involved)."""

    def __init__(self, n_images=8, H=120, W=160, device=None, seed=0):
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        rs = np.random.RandomState(seed)
        self.n_images, self.H, self.W = n_images, H, W
        f = 2.2 * W
        K = np.array([[f, 0, W / 2, 0], [0, f, H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
        poses = []
        for i in range(n_images):
            c = rs.standard_normal(3)
            c = c / np.linalg.norm(c) * 2.8
            z = -c / np.linalg.norm(c)
            x = np.cross(z, np.array([0.0, 0.0, 1.0]))
            x = x / (np.linalg.norm(x) + 1e-9)
            y = np.cross(z, x)
            pose = np.eye(4, dtype=np.float32)
            pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = x, y, z, c
            poses.append(pose)
        # images: the pixel rays of gen_rays_at (dataset.py:119-131) traced against the analytic scene
        Kinv = np.linalg.inv(K.astype(np.float64))[:3, :3]
        px, py = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="xy")
        pix = np.stack([px, py, np.ones_like(px)], -1).reshape(-1, 3)
        images, masks = [], []
        for pose in poses:
            v = pix @ Kinv.T
            v = v / np.linalg.norm(v, axis=-1, keepdims=True)
            d = v @ pose[:3, :3].astype(np.float64).T
            o = np.broadcast_to(pose[:3, 3].astype(np.float64), d.shape)
            rgb, m = render_scene(o, d)
            images.append(rgb.reshape(H, W, 3))
            masks.append(np.repeat(m.reshape(H, W, 1), 3, axis=2))
        self.images = torch.from_numpy(np.stack(images)).to(self.device)
        self.masks = torch.from_numpy(np.stack(masks)).to(self.device)
        self.intrinsics_all = torch.from_numpy(np.stack([K] * n_images)).to(self.device)
        self.intrinsics_all_inv = torch.inverse(self.intrinsics_all)
        self.pose_all = torch.from_numpy(np.stack(poses)).to(self.device)
        self.scale_mats_np = [np.eye(4, dtype=np.float32)] * n_images
        self.object_bbox_min = np.array([-1.01, -1.01, -1.01], dtype=np.float32)
        self.object_bbox_max = np.array([1.01, 1.01, 1.01], dtype=np.float32)


def export_dtu_scene(ds: "SyntheticDataset", out_dir: str, scale: float = 1.0, offset=(0.0, 0.0, 0.0)) -> str:
    """Write a SyntheticDataset as a DTU-format case (what `Dataset` and the reference's loader read, dataset.py:41-113):
    image/%03d.png, mask/%03d.png and cameras_sphere.npz with world_mat_i = K [R | t] (world coordinates = scale * unit
    sphere coordinates + offset) and scale_mat_i = the unit-sphere normalisation.  Channels are written so that a cv2-style
    read (BGR) returns ds.images."""
    from PIL import Image
    os.makedirs(os.path.join(out_dir, "image"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "mask"), exist_ok=True)
    cams = {}
    S = np.eye(4, dtype=np.float64)
    S[:3, :3] *= scale
    S[:3, 3] = np.asarray(offset, dtype=np.float64)
    for i in range(ds.n_images):
        img = (ds.images[i].cpu().numpy() * 256.0).clip(0, 255).astype(np.uint8)      # stored values are BGR / 256
        msk = (ds.masks[i].cpu().numpy() * 255.0).clip(0, 255).astype(np.uint8)
        Image.fromarray(img[..., ::-1].copy()).save(os.path.join(out_dir, "image", "%03d.png" % i))
        Image.fromarray(msk).save(os.path.join(out_dir, "mask", "%03d.png" % i))
        K = ds.intrinsics_all[i].cpu().numpy().astype(np.float64)
        pose = ds.pose_all[i].cpu().numpy().astype(np.float64)                        # camera-to-(unit sphere) world
        pose_w = S @ pose                                                              # camera-to-world, world = S * sphere
        pose_w[:3, :3] /= scale
        w2c = np.linalg.inv(pose_w)
        cams["world_mat_%d" % i] = K @ w2c
        cams["scale_mat_%d" % i] = S
    path = os.path.join(out_dir, "cameras_sphere.npz")
    np.savez(path, **cams)
    return out_dir
