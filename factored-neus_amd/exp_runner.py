#!/usr/bin/env python3
"""Stage-1 runner with the reference's command line (exp_runner.py:637-682) on the MI355X HIP backend.

    python exp_runner.py --mode train --conf ./confs/wmask.conf --case dtu_scan97 --type dtu [--is_continue] [--gpu 0]
    torchrun --nproc-per-node 8 exp_runner.py --mode train ...        # ray-sharded data parallel (RCCL)

Kept from the reference: flags, conf keys, the 4-term loss, Adam + warm-up/cosine schedule (exp_runner.py:229-238),
cos-anneal ratio (:223-227), checkpoint file names and dict keys (:266-278), so checkpoints interchange.
Not provided (outside the hot path): TensorBoard, video/relighting modes, the non-DTU loaders;
mesh extraction dumps the SDF grid (marching cubes needs PyMCubes, absent here).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from fneus import hocon, ops                      # noqa: E402
from fneus.parallel import init_from_env, broadcast_parameters     # noqa: E402
from fneus.trainer import Stage1Trainer           # noqa: E402
from models.dataset import Dataset, DatasetShiny, SyntheticDataset            # noqa: E402
from models.fields import NeRF                    # noqa: E402


class Runner:
    def __init__(self, conf_path, mode="train", case="CASE_NAME", is_continue=False, type="dtu", surface_weight=0.1,
                 device=None, prec=ops.PREC_PARITY, distributed=False, use_graph=True):
        self.device = device or torch.device("cuda")
        self.conf_path = conf_path
        self.conf = hocon.parse_file(conf_path, case)
        self.base_exp_dir = self.conf["general.base_exp_dir_geo"]
        os.makedirs(self.base_exp_dir, exist_ok=True)
        self.type, self.mode, self.surface_weight = type, mode, surface_weight
        if type == "dtu":
            self.dataset = Dataset(self.conf["dataset"], device=self.device)
        elif type == "synthetic":
            self.dataset = SyntheticDataset(device=self.device)
        elif type in ("shiny", "indisg_shiny"):          # reference exp_runner.py:50-51 / lvis.py:48-49
            self.dataset = DatasetShiny(self.conf["dataset"], device=self.device)
        else:
            raise NotImplementedError(f"--type {type}: 'dtu', 'shiny' and 'synthetic' feed the HIP hot path")
        tc = self.conf["train"]
        self.end_iter, self.save_freq, self.report_freq = tc.get_int("end_iter"), tc.get_int("save_freq"), tc.get_int("report_freq")
        self.val_freq, self.val_mesh_freq = tc.get_int("val_freq"), tc.get_int("val_mesh_freq")
        self.batch_size = tc.get_int("batch_size")
        self.validate_resolution_level = tc.get_int("validate_resolution_level")
        self.learning_rate, self.learning_rate_alpha = tc.get_float("learning_rate"), tc.get_float("learning_rate_alpha")
        self.use_white_bkgd = tc.get_bool("use_white_bkgd")
        self.warm_up_end, self.anneal_end = tc.get_float("warm_up_end", 0.0), tc.get_float("anneal_end", 0.0)
        self.igr_weight, self.mask_weight = tc.get_float("igr_weight"), tc.get_float("mask_weight")
        model_conf = {k: dict(self.conf["model"][k]) for k in ("sdf_network", "variance_network", "rendering_network",
                                                                "neus_renderer", "nerf")}
        self.trainer = Stage1Trainer(self.device, model_conf=model_conf, prec=prec, lr=self.learning_rate,
                                     igr_weight=self.igr_weight, mask_weight=self.mask_weight,
                                     surface_weight=surface_weight, synthetic_init=False, distributed=distributed,
                                     use_graph=use_graph)   # cos_anneal_ratio is a device scalar: a ramp replays too
        # constructed and checkpointed also when n_outside == 0 (never evaluated then), as in the reference
        self.nerf_outside = self.trainer.nerf_outside
        self.iter_step = 0
        if is_continue:
            names = sorted(n for n in os.listdir(os.path.join(self.base_exp_dir, "checkpoints"))
                           if n.endswith("pth") and int(n[5:-4]) <= self.end_iter)
            if names:
                logging.info("Find checkpoint: %s", names[-1])
                self.load_checkpoint(names[-1])
        broadcast_parameters(self.trainer.modules)

    # ---- schedules (exp_runner.py:223-238) ----
    def get_cos_anneal_ratio(self):
        return 1.0 if self.anneal_end == 0.0 else float(np.min([1.0, self.iter_step / self.anneal_end]))

    def update_learning_rate(self):
        if self.iter_step < self.warm_up_end:
            factor = self.iter_step / self.warm_up_end
        else:
            alpha = self.learning_rate_alpha
            progress = (self.iter_step - self.warm_up_end) / (self.end_iter - self.warm_up_end)
            factor = (np.cos(np.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha
        self.trainer.set_lr(self.learning_rate * factor)

    def train(self, max_steps=None, rank=0):
        self.update_learning_rate()
        perm = torch.randperm(self.dataset.n_images)
        bg = torch.ones([1, 3], device=self.device) if self.use_white_bkgd else None
        steps = self.end_iter - self.iter_step if max_steps is None else max_steps
        for _ in range(steps):
            data = self.dataset.gen_random_rays_at(perm[self.iter_step % len(perm)], self.batch_size)
            losses = self.trainer.train_step(data, cos_anneal_ratio=self.get_cos_anneal_ratio(), background_rgb=bg)
            self.iter_step += 1
            if self.iter_step % self.report_freq == 0:
                losses = self.trainer.global_losses(losses)      # data parallel: every rank joins this small all-reduce
            if rank == 0 and self.iter_step % self.report_freq == 0:
                print(self.base_exp_dir)
                print("iter:{:8>d} loss = {} lr={}".format(self.iter_step, losses["loss"].item(),
                                                           self.trainer.get_lr()))
            if rank == 0 and self.iter_step % self.save_freq == 0:
                self.save_checkpoint()
            self.update_learning_rate()
            if self.iter_step % len(perm) == 0:
                perm = torch.randperm(self.dataset.n_images)
        return losses

    # ---- checkpoints: same keys / file names as the reference (exp_runner.py:253-278) ----
    def save_checkpoint(self):
        t = self.trainer
        ckpt = {"nerf": self.nerf_outside.state_dict(), "sdf_network_fine": t.sdf_network.state_dict(),
                "variance_network_fine": t.deviation_network.state_dict(),
                "color_network_fine": t.color_network.state_dict(), "refColor_network": t.refColor_network.state_dict(),
                "optimizer": t.optimizer.state_dict(), "iter_step": self.iter_step}
        os.makedirs(os.path.join(self.base_exp_dir, "checkpoints"), exist_ok=True)
        torch.save(ckpt, os.path.join(self.base_exp_dir, "checkpoints", "ckpt_{:0>6d}.pth".format(self.iter_step)))

    def load_checkpoint(self, name):
        ckpt = torch.load(os.path.join(self.base_exp_dir, "checkpoints", name), map_location=self.device)
        t = self.trainer
        self.nerf_outside.load_state_dict(ckpt["nerf"])
        t.sdf_network.load_state_dict(ckpt["sdf_network_fine"])
        t.deviation_network.load_state_dict(ckpt["variance_network_fine"])
        t.color_network.load_state_dict(ckpt["color_network_fine"])
        t.refColor_network.load_state_dict(ckpt["refColor_network"])
        try:
            t.optimizer.load_state_dict(ckpt["optimizer"])
        except (ValueError, KeyError, RuntimeError):
            logging.warning("optimizer state of the checkpoint does not match this parameter list; starting Adam fresh")
        self.iter_step = ckpt["iter_step"]

    # ---- validation ----
    def validate_image(self, idx=-1, resolution_level=-1):
        """render one camera chunk by chunk (exp_runner.py:374-486) -> PNG via PIL (BGR like the reference).  The chunks are
        4096 rays (or batch_size if larger) and stay on the device until the image is complete: rays are independent, so
        the chunk size changes nothing but the number of launches and host synchronisations."""
        from PIL import Image
        if idx < 0:
            idx = np.random.randint(self.dataset.n_images)
        l = self.validate_resolution_level if resolution_level < 0 else resolution_level
        rays_o, rays_d = self.dataset.gen_rays_at(idx, resolution_level=l)
        H, W, _ = rays_o.shape
        out_rgb = []
        chunk = max(int(self.batch_size), 4096)
        for o, d in zip(rays_o.reshape(-1, 3).split(chunk), rays_d.reshape(-1, 3).split(chunk)):
            data = torch.cat([o, d, torch.zeros(len(o), 4, device=o.device)], -1)
            out = self.trainer.render_only(data, cos_anneal_ratio=self.get_cos_anneal_ratio())
            out_rgb.append(out["color_fine"].clone())       # (a replayed chunk graph hands out static buffers)
        img = (torch.cat(out_rgb, 0).reshape(H, W, 3) * 256).clip(0, 255).to(torch.uint8).cpu().numpy()
        os.makedirs(os.path.join(self.base_exp_dir, "validations_fine"), exist_ok=True)
        path = os.path.join(self.base_exp_dir, "validations_fine", "{:0>8d}_{}.png".format(self.iter_step, idx))
        Image.fromarray(img[..., ::-1]).save(path)
        return path

    def validate_mesh(self, world_space=False, resolution=512, threshold=0.0):
        """exp_runner.py:518-532: iso-surface of the SDF inside the object's bounding box -> meshes/{iter:0>8d}.ply
        (the grid through the HIP K1 kernel, the surface through models/mesh.py)"""
        from models.mesh import write_ply
        vertices, triangles = self.trainer.renderer.extract_geometry(self.dataset.object_bbox_min, self.dataset.object_bbox_max,
                                                                     resolution=resolution, threshold=threshold)
        os.makedirs(os.path.join(self.base_exp_dir, "meshes"), exist_ok=True)
        if world_space:
            sm = np.asarray(self.dataset.scale_mats_np[0])
            vertices = vertices * sm[0, 0] + sm[:3, 3][None]
        path = os.path.join(self.base_exp_dir, "meshes", "{:0>8d}.ply".format(self.iter_step))
        write_ply(path, vertices, triangles)
        logging.info("mesh: %d vertices, %d triangles -> %s", len(vertices), len(triangles), path)
        return path


def main():
    logging.basicConfig(level=logging.INFO, format="[%(filename)s:%(lineno)d] %(message)s")
    ap = argparse.ArgumentParser()
    ap.add_argument("--conf", type=str, default="./confs/wmask.conf")
    ap.add_argument("--mode", type=str, default="train")
    ap.add_argument("--mcube_threshold", type=float, default=0.0)
    ap.add_argument("--is_continue", default=False, action="store_true")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--case", type=str, default="")
    ap.add_argument("--type", type=str, default="dtu")
    ap.add_argument("--surface_weight", type=float, default=0.1)
    ap.add_argument("--idx", type=int, default=0)
    ap.add_argument("--prec", choices=["parity", "fast"], default="parity")
    ap.add_argument("--max_steps", type=int, default=None)
    ap.add_argument("--no_graph", action="store_true", help="launch every kernel of the step eagerly (no hipGraph replay)")
    args = ap.parse_args()
    # torchrun --nproc-per-node N: rays sharded by rank over RCCL (FNEUS_DIST_BACKEND=gloo: several ranks on ONE GPU, tests)
    rank, world, local = init_from_env(os.environ.get("FNEUS_DIST_BACKEND", "nccl"))
    gpu = local % max(torch.cuda.device_count(), 1) if world > 1 else args.gpu
    torch.cuda.set_device(gpu)
    runner = Runner(args.conf, args.mode, args.case, args.is_continue, args.type, args.surface_weight,
                    device=torch.device("cuda", gpu), prec=ops.PREC_PARITY if args.prec == "parity" else ops.PREC_FAST,
                    distributed=world > 1, use_graph=not args.no_graph)
    if args.mode == "train":
        if world > 1:
            torch.manual_seed(1234 + rank)           # every rank draws its own images, pixels and depth jitter (the networks
        runner.train(max_steps=args.max_steps, rank=rank)      # were built and broadcast before: same weights everywhere)
    elif args.mode == "validate_mesh":
        print(runner.validate_mesh(world_space=True, resolution=512, threshold=args.mcube_threshold))   # exp_runner.py:668
    elif args.mode == "validate_image":
        print(runner.validate_image(idx=args.idx))
    else:
        raise NotImplementedError(f"--mode {args.mode} is outside the stage-1 hot path")


if __name__ == "__main__":
    main()
