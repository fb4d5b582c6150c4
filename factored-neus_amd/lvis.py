#!/usr/bin/env python3
"""Stage-2 runner with the reference's command line (lvis.py:338-447) on the MI355X HIP backend.

    python lvis.py --mode train --conf ./confs/wmask.conf --case dtu_scan97 --type dtu [--is_continue] [--gpu 0]
    torchrun --nproc-per-node 8 lvis.py --mode train ...             # ray-sharded data parallel (RCCL)

Loads the newest stage-1 checkpoint of general.base_exp_dir_geo (lvis.py:94-102, 231-237), trains Lvis + IndirectLight
(train.lvis.*: 10 000 iterations x 512 rays, warm-up 1000 then cosine, lvis.py:205-215) and writes checkpoints with the
reference's keys (lvis.py:255-269) under general.base_exp_dir_lvis.  The validation image dumps (lvis.py:272-336) are
presentation and not provided.
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
from fneus.trainer2 import Stage2Trainer          # noqa: E402
from models.dataset import Dataset, DatasetShiny, SyntheticDataset            # noqa: E402


def _latest(ckpt_dir, end_iter):
    names = sorted(n for n in os.listdir(ckpt_dir) if n.endswith("pth") and int(n[5:-4]) <= end_iter)
    return names[-1] if names else None


class Runner:
    def __init__(self, conf_path, mode="train", case="CASE_NAME", is_continue=False, type="dtu", device=None,
                 prec=ops.PREC_PARITY, use_graph=True, distributed=False, rank=0):
        self.device = device or torch.device("cuda")
        self.conf = hocon.parse_file(conf_path, case)
        self.base_exp_dir_lvis = self.conf["general.base_exp_dir_lvis"]
        self.base_exp_dir_geometry = self.conf["general.base_exp_dir_geo"]
        os.makedirs(self.base_exp_dir_lvis, exist_ok=True)
        if type == "dtu":
            self.dataset = Dataset(self.conf["dataset"], device=self.device)
        elif type == "synthetic":
            self.dataset = SyntheticDataset(device=self.device)
        elif type in ("shiny", "indisg_shiny"):          # reference exp_runner.py:50-51 / lvis.py:48-49
            self.dataset = DatasetShiny(self.conf["dataset"], device=self.device)
        else:
            raise NotImplementedError(f"--type {type}: 'dtu', 'shiny' and 'synthetic' feed the HIP hot path")
        tc = self.conf["train"]
        self.end_iter = tc.get_int("lvis.end_iter")
        self.batch_size = tc.get_int("lvis.batch_size")
        self.warm_up_end = tc.get_float("lvis.warm_up_end", 0.0)
        self.save_freq, self.report_freq = tc.get_int("save_freq"), tc.get_int("report_freq")
        self.learning_rate, self.learning_rate_alpha = tc.get_float("learning_rate"), tc.get_float("learning_rate_alpha")
        model_conf = {k: dict(self.conf["model"][k]) for k in ("sdf_network", "variance_network", "rendering_network",
                                                                "lvis_renderer")}
        self.trainer = Stage2Trainer(self.device, model_conf=model_conf, prec=prec, lr=self.learning_rate, synthetic_init=False,
                                     use_graph=use_graph, distributed=distributed)
        self.rank = rank
        self.iter_step = 0
        geo = _latest(os.path.join(self.base_exp_dir_geometry, "checkpoints"), tc.get_int("end_iter"))
        if geo is None:
            raise FileNotFoundError(f"no stage-1 checkpoint under {self.base_exp_dir_geometry}/checkpoints")
        logging.info("Find geometry checkpoint: %s", geo)
        self.load_checkpoint_geometry(geo)
        if is_continue:
            name = _latest(os.path.join(self.base_exp_dir_lvis, "checkpoints"), self.end_iter)
            if name is not None:
                logging.info("Find checkpoint: %s", name)
                self.load_checkpoint(name)

    def _broadcast(self):
        """data parallel: every replica starts from rank 0's trainable parameters"""
        from fneus.parallel import broadcast_parameters
        t = self.trainer
        broadcast_parameters([t.lvis_network, t.indiLgt_network])

    def update_learning_rate(self):      # lvis.py:205-215
        if self.iter_step < self.warm_up_end:
            factor = self.iter_step / self.warm_up_end
        else:
            alpha = self.learning_rate_alpha
            progress = (self.iter_step - self.warm_up_end) / (self.end_iter - self.warm_up_end)
            factor = (np.cos(np.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha
        self.trainer.set_lr(self.learning_rate * factor)

    def train(self, max_steps=None):
        self._broadcast()
        self.update_learning_rate()
        perm = torch.randperm(self.dataset.n_images)
        steps = self.end_iter - self.iter_step if max_steps is None else max_steps
        losses = None
        for _ in range(steps):
            data = self.dataset.gen_random_rays_at(perm[self.iter_step % len(perm)], self.batch_size)
            out = self.trainer.train_step(data)
            if out is None:              # no ray hit the surface: the reference skips the batch WITHOUT counting it (lvis.py:160-161)
                continue
            losses = out
            self.iter_step += 1
            if self.rank == 0 and self.iter_step % self.report_freq == 0:
                print(self.base_exp_dir_lvis)
                print("iter:{:8>d} loss = {} lr={}".format(self.iter_step, losses["lvis_loss"].item(), self.trainer.get_lr()))
            if self.rank == 0 and self.iter_step % self.save_freq == 0:
                self.save_checkpoint()
            self.update_learning_rate()
            if self.iter_step % len(perm) == 0:
                perm = torch.randperm(self.dataset.n_images)
        return losses

    # ---- checkpoints: the reference's keys and file names (lvis.py:231-269) ----
    def _load_geometry(self, ckpt):
        t = self.trainer
        t.sdf_network.load_state_dict(ckpt["sdf_network_fine"])
        t.deviation_network.load_state_dict(ckpt["variance_network_fine"])
        t.color_network.load_state_dict(ckpt["color_network_fine"])
        t.refColor_network.load_state_dict(ckpt["refColor_network"])

    def load_checkpoint_geometry(self, name):
        self._load_geometry(torch.load(os.path.join(self.base_exp_dir_geometry, "checkpoints", name), map_location=self.device))

    def load_checkpoint(self, name):
        ckpt = torch.load(os.path.join(self.base_exp_dir_lvis, "checkpoints", name), map_location=self.device)
        self._load_geometry(ckpt)
        self.trainer.lvis_network.load_state_dict(ckpt["lvis_network"])
        self.trainer.indiLgt_network.load_state_dict(ckpt["indiLgt_network"])
        self.trainer.load_optimizer_state_dict(ckpt["optimizer"])
        self.iter_step = self.trainer.iter_step = ckpt["iter_step"]

    def save_checkpoint(self):
        t = self.trainer
        ckpt = {"sdf_network_fine": t.sdf_network.state_dict(), "variance_network_fine": t.deviation_network.state_dict(),
                "color_network_fine": t.color_network.state_dict(), "refColor_network": t.refColor_network.state_dict(),
                "lvis_network": t.lvis_network.state_dict(), "indiLgt_network": t.indiLgt_network.state_dict(),
                "optimizer": t.optimizer_state_dict(), "iter_step": self.iter_step}
        os.makedirs(os.path.join(self.base_exp_dir_lvis, "checkpoints"), exist_ok=True)
        torch.save(ckpt, os.path.join(self.base_exp_dir_lvis, "checkpoints", "ckpt_{:0>6d}.pth".format(self.iter_step)))


def main():
    logging.basicConfig(level=logging.INFO, format="[%(filename)s:%(lineno)d] %(message)s")
    ap = argparse.ArgumentParser()
    ap.add_argument("--conf", type=str, default="./confs/wmask.conf")
    ap.add_argument("--mode", type=str, default="train")
    ap.add_argument("--is_continue", default=False, action="store_true")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--case", type=str, default="")
    ap.add_argument("--type", type=str, default="dtu")
    ap.add_argument("--prec", choices=["parity", "fast"], default="parity")
    ap.add_argument("--max_steps", type=int, default=None)
    ap.add_argument("--no_graph", action="store_true", help="compact the hit points and launch eagerly instead of replaying the "
                                                             "fixed-shape step as a hipGraph")
    args = ap.parse_args()
    from fneus.parallel import init_from_env
    # torchrun --nproc-per-node N: rays sharded by rank over RCCL (FNEUS_DIST_BACKEND=gloo: several ranks on ONE GPU, tests)
    rank, world, local = init_from_env(os.environ.get("FNEUS_DIST_BACKEND", "nccl"))
    gpu = local % max(torch.cuda.device_count(), 1) if world > 1 else args.gpu
    torch.cuda.set_device(gpu)
    if world > 1:
        torch.manual_seed(1234 + rank)               # every rank draws its own pixels and directions
    runner = Runner(args.conf, args.mode, args.case, args.is_continue, args.type, device=torch.device("cuda", gpu),
                    prec=ops.PREC_PARITY if args.prec == "parity" else ops.PREC_FAST, use_graph=not args.no_graph,
                    distributed=world > 1, rank=rank)
    if args.mode == "train":
        runner.train(max_steps=args.max_steps)
    else:
        raise NotImplementedError(f"--mode {args.mode}: the validation image dumps of stage 2 are presentation, not provided")


if __name__ == "__main__":
    main()
