"""Stage-3 training step (mateIllu.py:135-203) on the HIP backend: material / illumination estimation.

Frozen inputs: the SDF, the RefColor head, Lvis and IndirectLight (mateIllu.py:83-89); trained: EnvmapMaterialNetwork
(128 light SGs, the BRDF auto-encoder, net_cs; mateIllu.py:91-95).  The step is
  mateIllu_render (renderer.py:630-726) -> masked L1 colour over the rays that hit + latent sparsity (mateIllu.py:152-172)
  -> backward -> Adam.
Cost: the primary sampler (K1 / K6) and the per-lobe light visibility, 128 lobes x 32 directions = 4096 Lvis evaluations per
hit point (<= 2.1 M per step).
"""
from __future__ import annotations

from typing import Optional

import torch

from fneus import ops, synth
from fneus.trainer import WMASK_MODEL


def stage3_loss(out: dict, true_rgb, mask, reduce=None):
    """mateIllu.py:152-172 (mask = (mask > 0.5) when train.mask_weight > 0, else ones: the caller's business).
    reduce(t) -> t summed over the data-parallel ranks: rgb_loss is then this rank's share of the global batch's"""
    if reduce is None and out["rgb"].is_cuda and out["rgb"].dtype == torch.float32 and mask.dtype == torch.float32:
        from fneus.autograd import Stage3LossFn                  # the sums below and their gradient in one launch (~30 before)
        rgb_loss, vals = Stage3LossFn.apply(out["rgb"], true_rgb.float(), mask, out["sdf_mask"])
        return {"loss": rgb_loss + out["encoder_loss"], "rgb_loss": rgb_loss, "encoder_loss": out["encoder_loss"], "psnr": vals[1]}
    w = mask * out["sdf_mask"][:, None].to(mask.dtype)          # rays that hit AND lie inside the image mask (no boolean
    wsum = w.sum().reshape(1)                                    # indexing: nothing here may synchronise with the host)
    if reduce is not None:
        wsum = reduce(wsum)
    denom = wsum[0] + 1e-5
    diff = out["rgb"] - true_rgb
    rgb_loss = (diff * w).abs().sum() / denom
    psnr = 20.0 * torch.log10(1.0 / ((diff ** 2 * w).sum() / (denom * 3.0)).sqrt())
    return {"loss": rgb_loss + out["encoder_loss"], "rgb_loss": rgb_loss, "encoder_loss": out["encoder_loss"], "psnr": psnr}


class Stage3Trainer:
    def __init__(self, device, model_conf: Optional[dict] = None, prec: int = ops.PREC_PARITY, lr: float = 5e-4, seed: int = 0,
                 synthetic_init: bool = True, mask_weight: float = 0.1, use_graph: bool = False, distributed: bool = False):
        from models.fields import SDFNetwork, SingleVarianceNetwork, RefColor, Lvis, IndirectLight
        from models.inverRender import EnvmapMaterialNetwork
        from models.renderer import NeuSRenderer
        conf = model_conf or WMASK_MODEL
        self.device = device
        self.sdf_network = SDFNetwork(**conf["sdf_network"])
        self.deviation_network = SingleVarianceNetwork(**conf["variance_network"])
        self.refColor_network = RefColor()
        self.lvis_network, self.indiLgt_network = Lvis(), IndirectLight()
        torch.manual_seed(seed)
        self.mateIllu_network = EnvmapMaterialNetwork()
        if synthetic_init:
            T = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
            self.sdf_network.load_state_dict(T(synth.sdf_state_dict(seed)))
            self.refColor_network.load_state_dict(T(synth.refcolor_state_dict(seed + 2)))
            self.lvis_network.load_state_dict(T(synth.lvis_state_dict(seed + 4)))
            self.indiLgt_network.load_state_dict(T(synth.indilgt_state_dict(seed + 5)))
            self.mateIllu_network.load_state_dict(T(synth.mateillu_state_dict(seed + 6)))
        self.frozen = [self.sdf_network, self.deviation_network, self.refColor_network, self.lvis_network, self.indiLgt_network]
        for m in self.frozen + [self.mateIllu_network]:
            m.to(device)
        for m in self.frozen:
            for p in m.parameters():
                p.requires_grad_(False)
        self.sdf_network.set_precision(prec)
        self.refColor_network.set_precision(prec)
        self.params = list(self.mateIllu_network.parameters())          # mateIllu.py:91-95
        self._init_step_mode(use_graph, lr, distributed)
        import os
        if self.grads is not None and os.environ.get("FNEUS_DIRECT_GRADS", "1") != "0":
            self._direct_modules = [self.mateIllu_network]      # its three MLPs: dW / db straight into the gradient arena
        self.mateIllu_network.stat_reduce = self.reduce          # global latent-sparsity statistics (data parallel)
        self.renderer = NeuSRenderer(**conf["neus_renderer"], sdf_network=self.sdf_network,
                                     deviation_network=self.deviation_network, refColor_network=self.refColor_network,
                                     lvis_network=self.lvis_network, indiLgt_network=self.indiLgt_network,
                                     mateIllu_network=self.mateIllu_network)
        self.mask_weight = mask_weight
        self.iter_step = 0

    # launch mode: see fneus/trainer2.py (fixed-shape step under one hipGraph)
    from fneus.trainer2 import Stage2Trainer as _S2
    _init_step_mode, set_lr, get_lr, _graph_step = _S2._init_step_mode, _S2.set_lr, _S2.get_lr, _S2._graph_step
    optimizer_state_dict, load_optimizer_state_dict = _S2.optimizer_state_dict, _S2.load_optimizer_state_dict
    _backward_and_step, _backward, _reduce, _clear_grads = _S2._backward_and_step, _S2._backward, _S2._reduce, _S2._clear_grads
    _direct_grads, _refresh_frozen = _S2._direct_grads, _S2._refresh_frozen
    del _S2

    def _fixed_shape_step(self, data: torch.Tensor):
        rays_o, rays_d, true_rgb, mask = ops.split_batch(data.contiguous())
        mask = (mask > 0.5).float() if self.mask_weight > 0.0 else torch.ones_like(mask)
        self._direct_grads(True)
        try:
            # (raw: fneus_stage3_loss skips the rays without a hit by their mask.  The data-parallel step's element-wise loss
            # MULTIPLIES by the mask: it keeps the placeholder rows filled with 1, a placeholder need not be finite)
            out = self.renderer.mateIllu_render(rays_o, rays_d, None, None, fixed_shape=True, keys=("rgb",), raw=self.reduce is None)
        finally:
            self._direct_grads(False)
        losses = stage3_loss(out, true_rgb, mask, self.reduce)
        self._backward_and_step(losses["loss"])
        return {"n_hit": out["sdf_mask"].sum(), **{k: v.detach() for k, v in losses.items()}}

    def train_step(self, data: torch.Tensor, near=None, far=None, u_theta=None, u_phi=None, z_vals_override=None):
        """data [B,10] (dataset.py:133-151).  -> loss dict, or None when no ray hits the surface (mateIllu.py:156)"""
        if self.use_graph and near is None and u_theta is None:
            return self._graph_step(data)
        if self.distributed:
            self.iter_step += 1
            return self._fixed_shape_step(data)
        rays_o, rays_d, true_rgb, mask = ops.split_batch(data.contiguous())
        mask = (mask > 0.5).float() if self.mask_weight > 0.0 else torch.ones_like(mask)
        out = self.renderer.mateIllu_render(rays_o, rays_d, near, far, u_theta=u_theta, u_phi=u_phi, z_vals_override=z_vals_override)
        if not bool(out["sdf_mask"].any()):
            return None
        losses = stage3_loss(out, true_rgb, mask)
        self._clear_grads()
        if self.grads is not None:
            self.grads.restore_small_grads()
        losses["loss"].backward()
        self.optimizer.step()
        self.iter_step += 1
        losses["n_hit"] = out["sdf_mask"].sum()
        return losses
