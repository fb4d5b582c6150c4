"""FlatAdam: torch.optim.Adam with its step() replaced by ONE fneus_adam launch over the whole model.

Same hyper-parameters, same per-parameter state (`step`, `exp_avg`, `exp_avg_sq`) and therefore the same
state_dict()/load_state_dict() format as the reference's optimiser (exp_runner.py:108, checkpoint key "optimizer"), but
* the moments of adjacent parameters live in one arena (the Parameters of a fused MLP are views of one flat buffer, so
  a whole network is a single contiguous segment for the kernel);
* `step` is one device scalar shared by every parameter, `lr` a device scalar: the step is hipGraph-capturable;
* the kernel clears each gradient after using it (zero_grad() becomes a no-op for the owner of this optimiser).
Every parameter that takes part must have a persistent `.grad` buffer when step() is first called (the trainer guarantees
it); a parameter whose `.grad` is None is skipped like torch.optim.Adam skips it (no state, no update): the reference hands
the never-evaluated background NeRF of the wmask configuration to Adam in exactly that way (exp_runner.py:89, 96).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib
from ._lib import lib, check


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clear_grads: bool = True):
        super().__init__(params, lr=float(lr), betas=betas, eps=eps)
        if len(self.param_groups) != 1:
            raise NotImplementedError("FlatAdam supports a single parameter group (as the reference uses)")
        self.clear_grads = clear_grads
        self._segs = None          # ctypes array of FneusAdamSegment
        self._keys = None          # (param ptr, grad ptr) per parameter at build time
        self._arenas = []
        self._step_dev = None
        self._lr_dev = None
        self._lr_host = None

    # ---- learning rate: a device scalar (graph replays see updates) ----
    def set_lr(self, lr: float):
        self.param_groups[0]["lr"] = float(lr)
        if self._lr_dev is not None:
            self._lr_dev.fill_(float(lr))
            self._lr_host = float(lr)

    def _params(self) -> List[torch.nn.Parameter]:
        """the parameters this step updates: those with a gradient (torch.optim.Adam's rule)"""
        return [p for p in self.param_groups[0]["params"] if p.requires_grad and p.grad is not None]

    def _build(self):
        ps = self._params()
        if not ps:
            raise RuntimeError("FlatAdam.step(): no parameter has a gradient")
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError("FlatAdam needs contiguous fp32 parameters with persistent .grad buffers")
        # the shared step counter: continue from a loaded state if there is one
        steps = [float(self.state[p]["step"]) for p in ps if p in self.state and "step" in self.state[p]]
        # (fneus_adam: float[2] = the step count and the launch's arrival counter; the state's `step` is the 0-d view of [0])
        self._step_buf = torch.zeros(2, dtype=torch.float32, device=dev)
        self._step_buf[0] = max(steps) if steps else 0.0
        self._step_dev = self._step_buf[0]
        self._lr_host = float(self.param_groups[0]["lr"])
        self._lr_dev = torch.full((), self._lr_host, dtype=torch.float32, device=dev)
        # merge parameters whose storage AND gradient storage are adjacent
        runs, cur = [], [ps[0]]
        for prev, p in zip(ps, ps[1:]):
            adj = (p.data_ptr() == prev.data_ptr() + prev.numel() * 4 and
                   p.grad.data_ptr() == prev.grad.data_ptr() + prev.numel() * 4)
            if adj:
                cur.append(p)
            else:
                runs.append(cur)
                cur = [p]
        runs.append(cur)
        self._arenas = []
        segs = (_lib.FneusAdamSegment * len(runs))()
        for i, run in enumerate(runs):
            count = sum(p.numel() for p in run)
            m = torch.zeros(count, dtype=torch.float32, device=dev)
            v = torch.zeros(count, dtype=torch.float32, device=dev)
            off = 0
            for p in run:
                st = self.state[p]
                mv, vv = m[off: off + p.numel()].view_as(p), v[off: off + p.numel()].view_as(p)
                if "exp_avg" in st:                       # loaded (or previously stepped) moments move into the arena
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"], st["step"] = mv, vv, self._step_dev
                off += p.numel()
            self._arenas.append((m, v))
            segs[i].param, segs[i].grad = run[0].data_ptr(), run[0].grad.data_ptr()
            segs[i].exp_avg, segs[i].exp_avg_sq, segs[i].count = m.data_ptr(), v.data_ptr(), count
        self._segs = segs
        self._keys = [(p.data_ptr(), p.grad.data_ptr()) for p in ps]

    def _stale(self):
        if self._segs is None:
            return True
        ps = self._params()
        if len(ps) != len(self._keys):
            return True
        for p, (pp, gp) in zip(ps, self._keys):
            st = self.state.get(p)
            if p.grad is None or p.data_ptr() != pp or p.grad.data_ptr() != gp or st is None or st.get("step") is not self._step_dev:
                return True
        return False

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdam.step() takes no closure")
        if self._stale():                    # first call, re-attached parameters, or a load_state_dict()
            self._build()
        lr = self.param_groups[0]["lr"]
        if float(lr) != self._lr_host:       # someone assigned param_groups[0]["lr"] directly (reference style)
            self.set_lr(float(lr))
        b1, b2 = self.param_groups[0]["betas"]
        check(lib.fneus_adam(self._segs, len(self._segs), C.c_void_p(self._lr_dev.data_ptr()),
                             C.c_void_p(self._step_dev.data_ptr()), float(b1), float(b2), float(self.param_groups[0]["eps"]),
                             int(self.clear_grads), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fneus_adam")

    def state_dict(self):
        """torch.optim.Adam's format, with an independent `step` tensor per parameter (inside this optimiser all
        parameters share one device scalar; a consumer such as torch.optim.Adam increments each entry separately)"""
        sd = super().state_dict()
        for st in sd["state"].values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone().cpu()
        return sd

    def load_state_dict(self, state_dict):
        """torch shares tensors of the right dtype / device with the dict it is given; the moments move into this
        optimiser's arenas at the next step(), so take private copies now (the source may keep training)"""
        # torch maps the saved state onto the parameters BY POSITION and checks only the count: a state_dict of another
        # parameter order would put, say, the background NeRF's moments on the SDF network.  Check the shapes first.
        saved = state_dict["state"]
        ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        ps = [p for g in self.param_groups for p in g["params"]]
        if len(ids) != len(ps):
            raise ValueError(f"optimizer state_dict holds {len(ids)} parameters, this optimiser {len(ps)}")
        for i, p in zip(ids, ps):
            st = saved.get(i)
            if st is not None and torch.is_tensor(st.get("exp_avg")) and tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of parameter {i} has shape {tuple(st['exp_avg'].shape)}, the parameter "
                                 f"{tuple(p.shape)}: the parameter lists differ in order")
        super().load_state_dict(state_dict)
        for st in self.state.values():
            for k in ("step", "exp_avg", "exp_avg_sq"):
                if torch.is_tensor(st.get(k)):
                    st[k] = st[k].clone()
        self._segs = None

    @property
    def n_segments(self):
        return 0 if self._segs is None else len(self._segs)
