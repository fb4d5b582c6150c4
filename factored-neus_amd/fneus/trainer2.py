"""Stage-2 training step (lvis.py:132-196) on the HIP backend: distil light visibility and indirect light.

The geometry and appearance networks of stage 1 are frozen (lvis.py:78-92 hands only Lvis + IndirectLight to Adam); the
step is   lvis_render (renderer.py:567-627)  ->  L1 visibility + L1 traced radiance (lvis.py:164-170)  ->  backward  ->  Adam.
Its cost is the ground truth: 4 secondary rays per visible surface point x 512 SDF samples on the K1 kernel (1.05 M points
for a 512-ray batch), K6 up-sampling, K2 on 65 k mid points, K2 + K4 at the hit points; the two distilled MLPs see at most
2048 / 512 rows.
"""
from __future__ import annotations

from typing import Optional

import torch

from fneus import ops, synth
from fneus.trainer import WMASK_MODEL

LVIS_RENDERER = dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)     # confs/wmask.conf:99-105


def stage2_loss(out: dict, reduce=None):
    """lvis.py:164-170.  reduce(t) -> t summed over the data-parallel ranks (in place): the loss terms are then this rank's
    SHARE of the global batch's (the hit count that normalises them is global), so that R ranks x B rays give the loss and the
    gradient of one R*B-ray batch when the shares / the rank gradients are summed"""
    m = out["sdf_mask"]
    if reduce is None and out["pre_lvis"].is_cuda and out["pre_lvis"].dtype == torch.float32 and out["pre_lvis"].dim() == 2 \
            and out["pre_lvis"].shape[1] == 4 and out["pre_lvis"].shape[0] == m.shape[0]:
        # both terms and their gradients in one launch (~35 element-wise ones).  Rows without a hit hold equal values on both
        # sides (1, or -- `raw` results of the fixed-shape step -- placeholders): the kernel skips them by the mask
        from fneus.autograd import Stage2LossFn
        loss, vals = Stage2LossFn.apply(out["pre_lvis"], out["pre_trace_radiance"], out["gt_lvis"], out["gt_trace_radiance"], m)
        return {"loss": loss, "lvis_loss": vals[0], "trace_radiance_loss": vals[1]}
    n_hit = m.sum().float().reshape(1)
    if reduce is not None:
        n_hit = reduce(n_hit)
    n_hit = n_hit[0]
    lvis_loss = (out["gt_lvis"] - out["pre_lvis"]).abs().sum() / (n_hit * 4 + 1e-6)
    err = (out["gt_trace_radiance"] - out["pre_trace_radiance"]) * m[:, None, None]
    radiance_loss = err.abs().sum() / (n_hit * 12 + 1e-6)
    return {"loss": lvis_loss + radiance_loss, "lvis_loss": lvis_loss, "trace_radiance_loss": radiance_loss}


class Stage2Trainer:
    def __init__(self, device, model_conf: Optional[dict] = None, prec: int = ops.PREC_PARITY, lr: float = 5e-4, seed: int = 0,
                 synthetic_init: bool = True, sdf_kwargs: Optional[dict] = None, use_graph: bool = False,
                 distributed: bool = False):
        from models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork, RefColor, Lvis, IndirectLight
        from models.renderer import NeuSRenderer
        conf = model_conf or dict(WMASK_MODEL, lvis_renderer=LVIS_RENDERER)
        self.device = device
        self.sdf_network = SDFNetwork(**conf["sdf_network"])
        self.color_network = RenderingNetwork(**conf["rendering_network"])
        self.deviation_network = SingleVarianceNetwork(**conf["variance_network"])
        self.refColor_network = RefColor()
        self.lvis_network, self.indiLgt_network = Lvis(), IndirectLight()
        if synthetic_init:
            T = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
            self.sdf_network.load_state_dict(T(synth.sdf_state_dict(seed, **(sdf_kwargs or {}))))
            self.color_network.load_state_dict(T(synth.color_state_dict(seed + 1)))
            self.refColor_network.load_state_dict(T(synth.refcolor_state_dict(seed + 2)))
            self.lvis_network.load_state_dict(T(synth.lvis_state_dict(seed + 4)))
            self.indiLgt_network.load_state_dict(T(synth.indilgt_state_dict(seed + 5)))
        self.frozen = [self.sdf_network, self.deviation_network, self.color_network, self.refColor_network]
        for m in self.frozen + [self.lvis_network, self.indiLgt_network]:
            m.to(device)
        for m in self.frozen:
            for p in m.parameters():
                p.requires_grad_(False)
        self.sdf_network.set_precision(prec)
        self.color_network.set_precision(prec)
        # lvis.py:89-92
        self.params = list(self.lvis_network.parameters()) + list(self.indiLgt_network.parameters())
        self._init_step_mode(use_graph, lr, distributed)
        self.renderer = NeuSRenderer(**conf.get("lvis_renderer", LVIS_RENDERER), sdf_network=self.sdf_network,
                                     deviation_network=self.deviation_network, color_network=self.color_network,
                                     lvis_network=self.lvis_network, indiLgt_network=self.indiLgt_network)
        self.iter_step = 0

    # ---- launch mode (shared with fneus/trainer3.py) -----------------------------------------------------------------------
    # use_graph: the step runs at FIXED SHAPE (every ray is treated as a hit point and masked afterwards: no host read of the
    # hit count, no data-dependent shape) and is replayed as ONE hipGraph after two eager warm-up steps -- the ~600
    # launch-bound element-wise kernels of the torch-side networks and losses then cost their GPU time only.  Differences a
    # user can observe: a batch without any hit is not skipped (its loss and gradients are zero; Adam's moments still
    # decay), the direction draws consume the generator per ray instead of per hit point.
    #
    # distributed (one process per GPU, rays sharded by rank): the fixed-shape step (eagerly, or as a
    # chain of hipGraphs cut at the collectives) with two exchanges -- the loss normalisers (hit count; stage 3: mask sum and the latent-sparsity statistics) summed over
    # the ranks BEFORE the loss, so that R ranks x B rays equal one R*B-ray batch, and ONE in-place all-reduce of the flat
    # gradient buffer all trained parameters' .grad are views of.  Every rank runs both every step (no data-dependent skip).
    def _init_step_mode(self, use_graph: bool, lr: float, distributed: bool = False):
        import os
        from fneus.seggraph import SegmentedStep
        # (the distilled MLPs of stages 2 / 3 run on the fneus_mlp_* kernels since round 5 -- models/fields.py seq_group: there is no
        # library GEMM left in a step, and the BLAS-backend preference rounds 3 / 4 set around every step is gone with it)
        self.graph_error = None             # why a graph capture fell back to eager launches, if it did
        self.distributed = bool(distributed)
        self.reduce = None
        self.grads = None
        self._seg = None
        if self.distributed:
            from fneus.parallel import GradArena
            # graphs + collectives: the step is recorded as a chain of hipGraphs cut at its collectives (fneus/seggraph.py);
            # FNEUS_DP_GRAPH=0 keeps data-parallel runs on eager launches
            use_graph = use_graph and os.environ.get("FNEUS_DP_GRAPH", "1") != "0"
            self._seg = SegmentedStep(self.device)
            self.reduce = self._reduce                            # in-place all-reduce(SUM); identity for one rank
        self.flat_adam = self.device.type == "cuda" and os.environ.get("FNEUS_STAGE_ADAM", "flat") == "flat"
        if self.distributed or self.flat_adam:                    # every trained parameter's .grad is a view of ONE flat buffer
            from fneus.parallel import GradArena
            holder = torch.nn.Module()
            holder.ps = torch.nn.ParameterList(self.params)
            self.grads = GradArena(self.device, [], None, [holder])
        if self.grads is not None and os.environ.get("FNEUS_DIRECT_GRADS", "1") != "0":
            # persistent, per-step cleared gradient buffers: the distilled MLPs write their weight gradients straight into them
            # (models/fields.py _DirectLinearFn: two launches fewer per layer than autograd's mm + accumulate)
            # -- only inside this trainer's own steps (_direct_grads_on / _off around the forward pass): there the buffers are
            # cleared before every backward; anyone else calling the modules gets autograd's accumulating route
            self._direct_modules = [m for m in (getattr(self, "lvis_network", None), getattr(self, "indiLgt_network", None))
                                    if m is not None and any(p.requires_grad for p in m.parameters())]
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        # FlatAdam (fneus/optim.py): torch.optim.Adam's state and state_dict with ONE fneus_adam launch per step, device-scalar
        # step counter and learning rate (graph-capturable), gradients cleared by the kernel.  MEASURED: torch's own step costs
        # 0.73 ms of GPU time per eager stage-2 step (20 tensors x its element-wise kernels), 0.5 ms inside the replayed graph.
        # FNEUS_STAGE_ADAM=torch keeps torch.optim.Adam.
        if self.flat_adam:
            from fneus.optim import FlatAdam
            self.optimizer = FlatAdam(self.params, lr=lr)
        elif self.use_graph:      # a device-scalar learning rate: schedule updates reach the replayed optimiser step
            self.optimizer = torch.optim.Adam(self.params, lr=torch.tensor(float(lr), device=self.device), capturable=True)
        else:
            self.optimizer = torch.optim.Adam(self.params, lr=lr)
        self._graph, self._eager_steps, self.graph_warmup_steps = None, 0, 2

    def _reduce(self, t: torch.Tensor) -> torch.Tensor:
        """loss normalisers summed over the ranks, in place; while a step is being recorded this is a cut of the graph chain"""
        from fneus.parallel import reduce_loss_norms
        self._seg.cut(lambda: reduce_loss_norms(t), t)
        return t

    def optimizer_state_dict(self):
        """torch.optim.Adam's format with host scalars (what the reference's checkpoints hold): the graph mode keeps `lr` and
        the step counters on the device"""
        sd = self.optimizer.state_dict()
        for g in sd["param_groups"]:
            if torch.is_tensor(g["lr"]):
                g["lr"] = float(g["lr"])
        for st in sd["state"].values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone().cpu()
        return sd

    def load_optimizer_state_dict(self, sd):
        self.optimizer.load_state_dict(sd)
        if self.use_graph:               # back to device scalars; a captured graph holds the old tensors: capture again
            if not self.flat_adam:
                for g in self.optimizer.param_groups:
                    g["lr"] = torch.tensor(float(g["lr"]), device=self.device)
            self._graph, self._eager_steps = None, 0

    def set_lr(self, lr: float):
        if self.flat_adam:
            self.optimizer.set_lr(float(lr))
            return
        for g in self.optimizer.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(float(lr))
            else:
                g["lr"] = lr

    def get_lr(self) -> float:
        return float(self.optimizer.param_groups[0]["lr"])

    def _graph_step(self, data: torch.Tensor):
        """fixed-shape step: eagerly while warming up, then captured once per batch shape and replayed.  Data parallel: the
        recording is a chain of graphs with the collectives between them (fneus/seggraph.py); if it fails on any rank,
        every rank stays on eager launches."""
        if self._graph is not None and self._graph[1].shape != data.shape:
            self._graph = None
        if self._graph is None:
            if self._eager_steps < self.graph_warmup_steps:
                self._eager_steps += 1
                self.iter_step += 1
                if self._seg is not None:
                    return self._seg.count_eagerly(lambda: self._fixed_shape_step(data))
                return self._fixed_shape_step(data)
            static = data.clone()
            if self._seg is not None:
                from fneus.parallel import collectives_active
                if not self._seg.record(lambda: self._fixed_shape_step(static), list(self._seg.shapes), collectives_active()):
                    self.use_graph = False
                    self.iter_step += 1
                    return self._fixed_shape_step(data)
                self._graph = (self._seg, static, self._seg.result)
                self._seg.replay()       # the recording pass only records: run the step for this batch now
                self.iter_step += 1
                return self._seg.result
            import gc
            gc.collect()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    losses = self._fixed_shape_step(static)
            except RuntimeError as e:   # a capture the runtime refuses must leave a working (eager) trainer behind; anything
                import sys              # else (a bug in the step) propagates
                self.graph_error = repr(e)
                print(f"[fneus] stage graph capture failed ({e!r}); continuing with eager launches", file=sys.stderr)
                torch.cuda.synchronize()
                self.use_graph = False
                self.iter_step += 1
                return self._fixed_shape_step(data)
            self._graph = (graph, static, losses)
            graph.replay()           # the capture pass only records: run the step for this batch now
            self.iter_step += 1
            return losses
        graph, static, losses = self._graph
        static.copy_(data)
        self._refresh_frozen()
        graph.replay()
        self.iter_step += 1
        return losses

    def _refresh_frozen(self):
        """A frozen network is packed when its parameters change (models/fields.py refresh: address + version), and the eager
        warm-up steps have packed it before the step was captured -- so the captured step holds no pack launch for it.  A
        load_state_dict (or any other version-bumping write) into a frozen network AFTER the capture is picked up here: its
        refresh() runs eagerly in front of the replay and packs into the same blob the captured kernels read (a no-op, one tuple
        comparison per network, while nothing changed)."""
        for m in getattr(self, "frozen", ()):
            for sub in m.modules():
                if hasattr(sub, "refresh") and not any(p.requires_grad for p in sub.parameters()):
                    sub.refresh()

    def _direct_grads(self, on: bool):
        for m in getattr(self, "_direct_modules", ()):
            m.direct_grads = on

    def _clear_grads(self):
        """torch.optim.Adam: drop the gradients; FlatAdam keeps persistent gradient buffers and clears them itself after use"""
        if not self.flat_adam:
            self.optimizer.zero_grad(set_to_none=True)

    def _backward(self, loss):
        """loss.backward() with a persistent unit seed: the fused loss nodes recognise it (ops.unit_loss_grad) and hand their saved
        gradients on without the multiplication by 1, and autograd does not fill a fresh seed every step"""
        one = getattr(self, "_seed_one", None)
        if one is None or one.device != loss.device or one.shape != loss.shape:
            one = self._seed_one = torch.ones_like(loss)
        with ops.unit_loss_grad(one):
            loss.backward(one)

    def _backward_and_step(self, loss):
        if self.grads is not None:           # gradients accumulate into the arena views (data parallel: summed in place)
            self.grads.restore_small_grads()
            if not self.flat_adam:           # (FlatAdam leaves them cleared)
                self.grads.flat.zero_()
            self._backward(loss)
            if self._seg is not None:
                self._seg.cut(self.grads.allreduce_sum, self.grads.flat)
        else:
            self._clear_grads()
            self._backward(loss)
        self.optimizer.step()

    def _fixed_shape_step(self, data: torch.Tensor):
        rays_o, rays_d, _rgb, _mask = ops.split_batch(data.contiguous())
        self._direct_grads(True)
        try:
            out = self.renderer.lvis_render(rays_o, rays_d, None, None, fixed_shape=True, raw=self.reduce is None)
        finally:
            self._direct_grads(False)
        losses = stage2_loss(out, self.reduce)
        self._backward_and_step(losses["loss"])
        return {"n_hit": out["sdf_mask"].sum(), **{k: v.detach() for k, v in losses.items()}}

    def train_step(self, data: torch.Tensor, near=None, far=None, u_theta=None, u_z=None, z_vals_override=None):
        """data [B,10] (dataset.py:133-151); near / far None: unit-sphere bounds.  -> loss dict, or None when no ray of
        the batch hits the surface (the reference skips such a batch, lvis.py:160-161)"""
        if self.use_graph and near is None and u_theta is None:
            return self._graph_step(data)
        if self.distributed:
            self.iter_step += 1
            return self._fixed_shape_step(data)
        rays_o, rays_d, _rgb, _mask = ops.split_batch(data.contiguous())
        out = self.renderer.lvis_render(rays_o, rays_d, near, far, u_theta=u_theta, u_z=u_z, z_vals_override=z_vals_override)
        if not bool(out["sdf_mask"].any()):
            return None
        losses = stage2_loss(out)
        self._clear_grads()
        if self.grads is not None:
            self.grads.restore_small_grads()
        losses["loss"].backward()
        self.optimizer.step()
        self.iter_step += 1
        losses["n_hit"] = out["sdf_mask"].sum()
        return losses
