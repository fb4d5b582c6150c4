"""Stage-1 training step on the HIP backend: render -> 4-term loss -> backward -> (gradient all-reduce) -> Adam.

Mirrors the body of the reference hot loop (exp_runner.py:131-181) without its per-step host synchronisations.
Used by exp_runner.py and bench.py.

With use_graph the whole step (about 20 fneus kernels and ~500 small PyTorch kernels of the RefColor head, the losses
and the optimiser) is captured once into a hipGraph and replayed: every shape on the path is static (fixed ray batch,
fixed-size sampler, no data-dependent indexing), and at ~6 ms per step the ~1.5 ms of launch gaps were the largest
single item left outside the fused kernels.
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from fneus import ops, synth
from fneus.optim import FlatAdam
from fneus.parallel import GradArena, collectives_active, reduce_loss_norms

WMASK_MODEL = {   # confs/wmask.conf:49-97
    "sdf_network": dict(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                        geometric_init=True, weight_norm=True),
    "variance_network": dict(init_val=0.3),
    "rendering_network": dict(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=4, weight_norm=True,
                              multires_view=4, squeeze_out=True),
    "neus_renderer": dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0),
}


class Stage1Trainer:
    def __init__(self, device, model_conf: Optional[dict] = None, prec: int = ops.PREC_PARITY, lr: float = 5e-4,
                 igr_weight: float = 0.1, mask_weight: float = 0.1, surface_weight: float = 0.1, seed: int = 0,
                 synthetic_init: bool = True, distributed: bool = False, use_graph: bool = False, gprec=None):
        from models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork, RefColor, NeRF
        from models.renderer import NeuSRenderer
        conf = model_conf or WMASK_MODEL
        self.device = device
        self.sdf_network = SDFNetwork(**conf["sdf_network"])
        self.color_network = RenderingNetwork(**conf["rendering_network"])
        self.deviation_network = SingleVarianceNetwork(**conf["variance_network"])
        self.refColor_network = RefColor()
        if synthetic_init:   # deterministic numpy-stream weights (same on every box), reference distributions
            T = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
            self.sdf_network.load_state_dict(T(synth.sdf_state_dict(seed)))
            self.color_network.load_state_dict(T(synth.color_state_dict(seed + 1)))
            self.refColor_network.load_state_dict(T(synth.refcolor_state_dict(seed + 2)))
        # The background NeRF++ is constructed, checkpointed and handed to Adam whether or not it is evaluated, exactly as
        # the reference does (exp_runner.py:82, 89, 96): with n_outside == 0 its 24 parameters never receive a gradient
        # and Adam skips them, but they hold positions 0..23 of the optimiser's parameter list -- which is what makes
        # optimiser state_dicts interchangeable with the reference's.
        self.use_nerf = conf["neus_renderer"].get("n_outside", 0) > 0      # womask: renderer.py:452-458
        self.nerf_outside = NeRF(**conf.get("nerf", dict(D=8, d_in=4, d_in_view=3, W=256, multires=10,
                                                         multires_view=4, output_ch=4, skips=[4], use_viewdirs=True)))
        if synthetic_init:
            self.nerf_outside.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(seed + 3).items()})
        # reference order (exp_runner.py:89-94): nerf_outside, sdf_network, deviation_network, color_network, refColor_network
        self.modules = [self.nerf_outside, self.sdf_network, self.deviation_network, self.color_network, self.refColor_network]
        for m in self.modules:
            m.to(device)
        self.sdf_network.set_precision(prec)
        for m in (self.sdf_network, self.color_network, self.refColor_network, self.nerf_outside):
            m.set_gradient_precision(gprec)                 # None: ops.DEFAULT_GPREC (bf16 planes)
        self.color_network.set_precision(prec)
        self.refColor_network.set_precision(prec)
        self.nerf_outside.set_precision(prec)
        self.params = [p for m in self.modules for p in m.parameters()]
        # data parallel: the collectives sit inside the step (the 4-float all-reduce of the loss normalisers before the
        # loss; the gradient arena in two parts, the early one beside the SDF backward, the late one after it), so the step
        # is captured as FOUR hipGraphs (three with FNEUS_DP_EARLY=0: one arena exchange) with the collectives launched
        # eagerly between their replays (_capture_dp).  FNEUS_DP_GRAPH=0 keeps such runs on eager launches.
        import os
        self.distributed = bool(distributed)
        self.use_graph = bool(use_graph) and device.type == "cuda" and (
            not distributed or os.environ.get("FNEUS_DP_GRAPH", "1") != "0")
        self._capturing = None
        self._in_dp_step = False
        self.reduce_norms = self._reduce_norms_hook if distributed else None
        # every gradient of the model lives in ONE arena: the fused MLPs accumulate into slices of it, the small torch
        # modules get persistent .grad views (autograd accumulates into them in place).  FlatAdam addresses parameters
        # and gradients by pointer and clears the gradients itself; data parallel = one in-place all-reduce of the arena.
        # arena order: [SDF | background NeRF] = the LATE part (final only when the backward ends), then colour network,
        # RefColor heads, variance = the EARLY part (final when the SDF backward starts; exchanged beside it)
        late = [self.sdf_network] + ([self.nerf_outside] if self.use_nerf else [])
        self.grads = GradArena(device, late + [self.color_network], self.refColor_network, [self.deviation_network],
                               n_late=len(late))
        # FNEUS_DP_EARLY = 1 / 0 fixes the form (split / single exchange); "auto" (the default of bench.py and the runners with
        # more than one rank, through autotune_exchange) measures both on the job's own steps and keeps the faster one
        self._warm_keys = set()
        self.split_exchange = self.distributed and os.environ.get("FNEUS_DP_EARLY", "1") != "0"
        self.exchange_choice = None          # filled by autotune_exchange: {"choice", "ms_split", "ms_single", "steps"}
        self._xstream = torch.cuda.Stream(device=device) if (self.distributed and device.type == "cuda") else None
        self._early = None           # handle of the early part's all-reduce of the step in flight
        if self.distributed:         # (the hook looks at split_exchange when it runs: both forms can be recorded by one trainer)
            self.sdf_network._ws.pre_backward = self._early_exchange
            # (the split exchange ships the colour network's gradients when the SDF backward starts: its weight-gradient
            # products must not wait for the SDF network's launch then, fneus/autograd.py ColorFn.backward)
            self.sdf_network._ws.color_grads_early = lambda: self.split_exchange
        self.optimizer = FlatAdam(self.params, lr=lr)
        self._graphs = {}            # (batch shape, background shape) -> (graph(s), static input, static background, losses)
        self._cos = torch.ones(1, dtype=torch.float32, device=device)    # cos_anneal_ratio of the replayed step
        self._cos_val = 1.0
        self._eager_steps = 0
        self.graph_warmup_steps = 2  # eager steps before the first capture (workspaces, job tables, LDS attributes)
        self.renderer = NeuSRenderer(**conf["neus_renderer"], nerf=self.nerf_outside if self.use_nerf else None,
                                     sdf_network=self.sdf_network,
                                     deviation_network=self.deviation_network, color_network=self.color_network,
                                     refColor_network=self.refColor_network)
        self.igr_weight, self.mask_weight, self.surface_weight = igr_weight, mask_weight, surface_weight
        self.bucket = self.grads if distributed else None
        self.iter_step = 0
        self._auto = None
        self._auto_begin()

    def set_lr(self, lr: float):
        self.optimizer.set_lr(lr)

    def get_lr(self) -> float:
        return float(self.optimizer.param_groups[0]["lr"])

    def train_step(self, data: torch.Tensor, cos_anneal_ratio: float = 1.0, background_rgb=None):
        """data [B,10] = rays_o, rays_d, rgb, mask (dataset.py:133-151).  Returns the loss dict (device tensors; with
        use_graph they are static buffers that the next step overwrites)."""
        if self._auto is not None:
            self._auto_advance()
        if not self.use_graph or ops.PROFILE is not None:
            return self._eager_step(data, cos_anneal_ratio, background_rgb)
        # one capture per batch shape: cos_anneal_ratio is a device scalar that the compositing kernels read at run time
        # (it ramps every step in the womask configuration), the background colour a static buffer
        key = (tuple(data.shape), None if background_rgb is None else tuple(background_rgb.shape), self.split_exchange)
        entry = self._graphs.get(key)
        if entry is None:
            # (one eager step per form of the step as well: the two forms of the gradient exchange use different job tables of the
            # weight-gradient GEMM, and a table is built -- host to device -- by the first step that needs it)
            if self._eager_steps < self.graph_warmup_steps or len(self._graphs) >= 6 or key not in self._warm_keys:
                self._warm_keys.add(key)
                return self._eager_step(data, cos_anneal_ratio, background_rgb)
            entry = (self._capture_dp if self.distributed else self._capture)(data, background_rgb)
            self._graphs[key] = entry
            if entry is None:            # capture failed (data parallel only): this and every later step runs eagerly
                return self._eager_step(data, cos_anneal_ratio, background_rgb)
        graph, static_data, static_bg, losses = entry
        static_data.copy_(data)
        if float(cos_anneal_ratio) != self._cos_val:        # (constant in wmask.conf: no launch per step)
            self._cos_val = float(cos_anneal_ratio)
            self._cos.fill_(self._cos_val)
        if static_bg is not None:
            static_bg.copy_(background_rgb)
        if self.distributed:
            g1, g2, g2b, g3, norms = graph
            g1.replay()                      # packs, sampler, K2, colour, compositing, surface gather, RefColor, batch sums
            reduce_loss_norms(norms)         # in place on the static buffer that the loss kernel of g2 reads
            g2.replay()                      # losses + the backward (with the split exchange: up to the SDF backward)
            if g2b is not None:
                early = self.grads.allreduce_early(self._xstream)     # colour / RefColor / variance gradients: beside ...
                g2b.replay()                 # ... the SDF backward (K3, its weight-gradient GEMM, weight-norm backward)
                self.grads.allreduce_late()
                self.grads.wait_early(early)
            else:
                self.bucket.allreduce_sum()
            g3.replay()                      # Adam
        else:
            graph.replay()
        self.iter_step += 1
        return losses

    def _reduce_norms_hook(self, norms: torch.Tensor) -> torch.Tensor:
        """called by the fused loss (Stage1LossFn) with this rank's batch sums; while _capture_dp records a step this is
        the point where the first graph ends and the second begins"""
        st = self._capturing
        if st is None:
            return reduce_loss_norms(norms)
        st["g1"].capture_end()
        st["open"] = None
        st["norms"] = norms
        reduce_loss_norms(norms)             # eager: nothing recorded has run yet, the values are meaningless, but every
        st["n_coll"] += 1                    # rank issues the same collectives
        # with the split exchange this graph is ended by _early_exchange ON THE AUTOGRAD ENGINE'S THREAD (the hook runs
        # inside the backward): only a relaxed-mode capture may be ended from another thread
        st["g2"].capture_begin(pool=st["pool"], capture_error_mode="relaxed" if self.split_exchange else "thread_local")
        st["open"] = st["g2"]
        return norms

    def _early_exchange(self):
        """called when the SDF backward starts (fneus/autograd.py SdfValueGradFn.backward): the gradients of every other
        network are final.  Eager step: start their all-reduce on the exchange stream.  While _capture_dp records a step:
        the second graph ends here and the third (the SDF backward) begins."""
        if not self.split_exchange:          # single exchange behind the backward: nothing happens here
            return
        st = self._capturing
        # FNEUS_OVERLAP bits 2 | 4 put the colour / RefColor weight-gradient GEMMs and the fold backward on the ops side
        # stream: they write the early part of the arena, so that stream joins before the exchange reads it (during a
        # capture the join also closes the fork, without which capture_end fails)
        ops.overlap_join()
        ops.flush_wn_batch()                 # the fold backward of the colour network and the RefColor heads: their gradients are read next
        if st is None:
            if self._in_dp_step:
                self._early = self.grads.allreduce_early(self._xstream)
            return
        st["g2"].capture_end()
        st["open"] = None
        h = self.grads.allreduce_early(self._xstream)      # eager, on meaningless values (see _capture_dp)
        self.grads.wait_early(h)
        st["n_coll"] += 1
        st["g2b"].capture_begin(pool=st["pool"], capture_error_mode="relaxed")    # begun on the autograd thread, ended on the caller's
        st["open"] = st["g2b"]

    def _capture_dp(self, data: torch.Tensor, background_rgb):
        """data parallel: four graphs per step with the three collectives between them (three and two with
        FNEUS_DP_EARLY=0).  Returns None when the capture fails ON ANY RANK (every rank then stays on eager launches).  The
        outcome is collective: whatever happens, every rank issues exactly the same collectives here -- the 4-float
        normaliser exchange, the arena all-reduce(s) (all on meaningless values: nothing recorded has run) and a MIN
        all-reduce of its success flag -- so a rank whose capture throws cannot pair its first real collectives with its
        peers' dummy ones."""
        import gc
        static_data = data.clone()
        static_bg = None if background_rgb is None else background_rgb.clone()
        gc.collect()
        torch.cuda.synchronize()
        g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        g2b = torch.cuda.CUDAGraph() if self.split_exchange else None
        st = {"g1": g1, "g2": g2, "g2b": g2b, "pool": torch.cuda.graph_pool_handle(), "open": None, "norms": None, "n_coll": 0}
        last = g2b if self.split_exchange else g2      # the graph that must be open when the backward has been recorded
        n_expected = 3 if self.split_exchange else 2
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ok, losses = False, None
        try:
            with torch.cuda.stream(side):
                # thread_local: the process group's watchdog thread may query events while this thread records
                g1.capture_begin(pool=st["pool"], capture_error_mode="thread_local")
                st["open"] = g1
                self._capturing = st
                losses = self._step_body(static_data, self._cos, static_bg, with_optimizer=False)
                self._capturing = None
                if st["open"] is not last:
                    raise RuntimeError("the step did not reach the loss-normaliser exchange" if st["open"] is g1 else
                                       "the step did not reach the SDF backward")
                last.capture_end()
                st["open"] = None
                if self.split_exchange:              # eager, like the exchanges above
                    self.grads.allreduce_late()
                else:
                    self.bucket.allreduce_sum()
                st["n_coll"] += 1
                g3.capture_begin(pool=st["pool"], capture_error_mode="thread_local")
                st["open"] = g3
                self.optimizer.step()
                g3.capture_end()
                st["open"] = None
            ok = True
        except Exception as e:      # noqa: BLE001 -- any failure here must leave a working (eager) trainer behind
            import sys
            print(f"[fneus] data-parallel graph capture failed ({e!r}); continuing with eager launches", file=sys.stderr)
        finally:
            self._capturing = None
            if st["open"] is not None:       # a capture is still open: end it ON THE STREAM THAT IS CAPTURING, or that stream
                try:                         # stays in capture mode and the next synchronisation fails
                    with torch.cuda.stream(side):
                        st["open"].capture_end()
                except Exception:   # noqa: BLE001
                    pass
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        # make up for the collectives a failed capture did not reach, then agree on the outcome
        import torch.distributed as dist
        if st["n_coll"] < 1:
            reduce_loss_norms(torch.zeros(4, dtype=torch.float32, device=self.device))
        if self.split_exchange:
            if st["n_coll"] < 2:
                self.grads.wait_early(self.grads.allreduce_early(self._xstream))
            if st["n_coll"] < 3:
                self.grads.allreduce_late()
        elif st["n_coll"] < 2:
            self.bucket.allreduce_sum()
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=self.device)
        if collectives_active():
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        if flag.item() < 0.5:                # some rank failed: ALL ranks run eagerly from here on
            if ok:
                import sys
                print("[fneus] data-parallel graph capture failed on another rank; continuing with eager launches", file=sys.stderr)
            self.use_graph = False
            self.grads.flat.zero_()          # (the dummy exchanges may have summed a failing rank's partial gradients)
            return None
        return (g1, g2, g2b, g3, st["norms"]), static_data, static_bg, losses

    # ---- which form of the gradient exchange?  ("overlap or direct reduce-scatter -- measure both") -----------
    # Data parallel with more than one rank and FNEUS_DP_EARLY unset (or "auto"): the job measures both forms on ITS OWN training
    # steps -- warm-up / capture steps, then AUTO_STEPS timed steps with the arena exchanged in two parts (four hipGraphs around
    # three collectives), the same with one exchange behind the backward (three graphs around two) -- takes the MAX over the
    # ranks of each wall time (one small all-reduce: every rank sees the same two numbers, so every rank takes the same
    # decision) and keeps the faster form.  The steps are ordinary training steps: nothing is repeated or thrown away.
    AUTO_STEPS = 20

    def _auto_begin(self):
        import os
        forced = os.environ.get("FNEUS_DP_EARLY", "auto")
        if not self.distributed or not collectives_active() or forced in ("0", "1"):
            self.exchange_choice = {"choice": "split" if self.split_exchange else "single", "ms_split": None, "ms_single": None,
                                    "steps": 0, "why": ("FNEUS_DP_EARLY=" + forced) if forced in ("0", "1") else "no collectives"}
            self._auto = None
            return
        self.split_exchange = True
        self._auto = {"phase": 0, "count": 0, "t0": 0.0, "ms": {}}

    def _auto_advance(self):
        """called at the start of every train_step while the selection runs"""
        import time
        import torch.distributed as dist
        a = self._auto
        warm = self.graph_warmup_steps + 3                           # eager warm-up where needed, the capture, first replays
        if a["phase"] in (0, 2) and a["count"] == warm:              # warm-up of this form done: start its clock
            torch.cuda.synchronize()
            dist.barrier()
            a["t0"], a["count"], a["phase"] = time.perf_counter(), 0, a["phase"] + 1
        elif a["phase"] in (1, 3) and a["count"] == self.AUTO_STEPS:
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - a["t0"]) / self.AUTO_STEPS * 1e3], dtype=torch.float32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            a["ms"]["split" if a["phase"] == 1 else "single"] = float(t.item())
            if a["phase"] == 1:
                self.split_exchange, a["phase"], a["count"] = False, 2, 0
            else:
                ms = a["ms"]
                self.split_exchange = ms["split"] <= ms["single"]
                self.exchange_choice = {"choice": "split" if self.split_exchange else "single", "ms_split": round(ms["split"], 4),
                                        "ms_single": round(ms["single"], 4), "steps": self.AUTO_STEPS,
                                        "why": "measured on this job's own steps (max over ranks)"}
                self._auto = None
                if int(os.environ.get("RANK", "0")) == 0 or os.environ.get("FNEUS_DP_VERBOSE"):
                    import sys
                    c = self.exchange_choice
                    print(f"[fneus] gradient exchange: {c['choice']} (split {c['ms_split']} ms, single {c['ms_single']} ms per step "
                          f"over {c['steps']} steps each, max over ranks)", file=sys.stderr, flush=True)
                return
        a["count"] += 1

    def autotune_exchange(self, batches, cos_anneal_ratio: float = 1.0, background_rgb=None):
        """run training steps on `batches` until the selection above has finished (bench.py: ahead of its warm-up, so that no timed
        step contains a capture); returns the record {"choice", "ms_split", "ms_single", "steps", "why"}"""
        i = 0
        while self._auto is not None:
            self.train_step(batches[i % len(batches)], cos_anneal_ratio, background_rgb)
            i += 1
        return self.exchange_choice

    def _capture(self, data: torch.Tensor, background_rgb):
        import gc
        static_data = data.clone()
        static_bg = None if background_rgb is None else background_rgb.clone()
        gc.collect()                 # drop autograd graphs of earlier eager steps that are only kept alive by cycles
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            losses = self._step_body(static_data, self._cos, static_bg, with_optimizer=True)
        return graph, static_data, static_bg, losses

    def _eager_step(self, data, cos_anneal_ratio, background_rgb):
        self._in_dp_step, self._early = self.split_exchange, None
        losses = self._step_body(data, cos_anneal_ratio, background_rgb, with_optimizer=False)
        self._in_dp_step = False
        if self.split_exchange:              # the early part has been on its way since the SDF backward started
            self.grads.allreduce_late()
            self.grads.wait_early(self._early)
            self._early = None
        elif self.bucket is not None:
            self.bucket.allreduce_sum()      # losses are normalised by the GLOBAL batch: the rank gradients just add up
        self.optimizer.step()
        self.iter_step += 1
        self._eager_steps += 1
        return losses

    def _step_body(self, data, cos_anneal_ratio, background_rgb, with_optimizer: bool):
        if data.is_cuda and data.dtype == torch.float32 and data.is_contiguous() and data.shape[1] == 10:
            rays_o, rays_d, true_rgb, mask = ops.split_batch(data)          # one launch instead of four strided copies
        else:
            rays_o, rays_d, true_rgb, mask = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
        ops.overlap_begin(self.device)       # window for side-stream work (fneus/ops.py); joined below, before Adam
        # near / far = None: near_far_from_sphere (dataset.py:186-192) is evaluated inside render's ray set-up launch
        # the losses of exp_runner.py:141-177 are evaluated inside render (fused with the surface shading and their own
        # gradients: one launch instead of ~200 element-wise kernels on [B]-ray tensors)
        out = self.renderer.render(rays_o, rays_d, None, None, background_rgb=background_rgb,
                                   cos_anneal_ratio=cos_anneal_ratio,
                                   loss_args=(true_rgb, mask, self.igr_weight, self.mask_weight, self.surface_weight,
                                              self.reduce_norms))
        losses = out["losses"]
        self.zero_grad()
        # (the seed of the backward pass is a constant 1 kept on the device: `backward()` alone fills a fresh one every step)
        one = self.__dict__.get("_seed_one")
        if one is None or one.device != losses["loss"].device or one.shape != losses["loss"].shape:
            one = self._seed_one = torch.ones_like(losses["loss"])
        if not ops.OVERLAP_MASK:     # the fold-backward launches of the networks as one (data parallel: one for the early part of
            with ops.batched_wn_backward(), ops.unit_loss_grad(one):     # the arena, flushed by _early_exchange, and the SDF
                losses["loss"].backward(one)                          # network's behind the backward)
        else:
            with ops.unit_loss_grad(one):
                losses["loss"].backward(one)
        ops.overlap_end()                    # the weight gradients issued on the side stream are complete from here on
        if with_optimizer:
            self.optimizer.step()
        # Hand out DETACHED values: a caller that keeps the loss of the previous step (for logging) would otherwise keep
        # that step's autograd graph -- and its AccumulateGrad nodes, bound to the stream they were created on -- alive
        # into the graph capture, which then records cross-stream work and crashes hipStreamEndCapture.
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()}

    def global_losses(self, losses: dict) -> dict:
        """data parallel: the loss terms of a step are this rank's SHARE of the global batch's; sum them over the ranks
        (one small all-reduce; call it on every rank, e.g. only when logging)"""
        import torch.distributed as dist
        if self.bucket is None or not collectives_active():
            return losses
        keys = ["loss", "color_loss", "surface_loss", "eikonal_loss", "mask_loss"]
        v = torch.stack([losses[k].detach().reshape(()) for k in keys])
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        out = dict(losses)
        for i, k in enumerate(keys):
            out[k] = v[i]
        return out

    def zero_grad(self):
        """nothing to launch: FlatAdam clears every gradient in its own pass, fneus_wn_backward the effective-gradient
        buffers; only gradients that something else set to None get their arena view back"""
        self.grads.restore_small_grads()

    def render_only(self, data: torch.Tensor, cos_anneal_ratio: float = 1.0):
        """forward-only render of a ray chunk (exp_runner.py:374-486 validate_image, per chunk).  With use_graph the ~25 launches
        of a chunk shape are captured once and replayed (FNEUS_RENDER_GRAPH=0: eager): the returned tensors are then static
        buffers that the next call of the same shape overwrites -- clone what has to outlive it."""
        import os
        if self.use_graph and data.is_cuda and ops.PROFILE is None and os.environ.get("FNEUS_RENDER_GRAPH", "1") != "0":
            cache = self.__dict__.setdefault("_render_graphs", {})
            # what a capture bakes in as host constants besides the shape: sample counts, precision modes, the modules' backends
            r = self.renderer
            key = (tuple(data.shape), data.dtype, r.n_samples, r.n_importance, r.n_outside, r.up_sample_steps,
                   getattr(self.sdf_network, "prec", None), getattr(self.color_network, "prec", None), ops.DEFAULT_GPREC)
            ent = cache.get(key)
            if ent is None and len(cache) >= 6:          # (bounded like the step graphs: a graph holds its memory pool)
                cache.pop(next(iter(cache)))
            if ent is None:
                self._render_eager(data, cos_anneal_ratio)          # allocations and lazy set-up outside the capture
                static = data.clone()
                cos = torch.full((1,), float(cos_anneal_ratio), dtype=torch.float32, device=data.device)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self._render_eager(static, cos)
                ent = cache[key] = (graph, static, cos, out)
            graph, static, cos, out = ent
            static.copy_(data)
            cos.fill_(float(cos_anneal_ratio))
            graph.replay()
            return out
        return self._render_eager(data, cos_anneal_ratio)

    def _render_eager(self, data: torch.Tensor, cos_anneal_ratio):
        rays_o, rays_d = data[:, :3], data[:, 3:6]
        with torch.no_grad():
            return self.renderer.render(rays_o, rays_d, None, None, perturb_overwrite=0, cos_anneal_ratio=cos_anneal_ratio)


def synthetic_batches(n_batches: int, batch: int, device, seed0: int = 1000, rank: int = 0):
    """DTU-shaped batches, one synthetic camera per step; resident on the device."""
    out = []
    for i in range(n_batches):
        out.append(torch.from_numpy(synth.ray_batch(batch, seed=seed0 + 7919 * rank + i)).to(device))
    return out
