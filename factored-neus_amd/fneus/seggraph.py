"""A training step recorded as a CHAIN of hipGraphs that is cut at every collective.

RCCL collectives are not recorded: a data-parallel step that replays graphs has them between the replays.  SegmentedStep
records `body()` once; wherever the body calls `cut(fn, like)` the running graph ends, `fn()` (an in-place collective on a
tensor that lives in the graphs' shared memory pool) runs eagerly -- on meaningless values, nothing recorded has run yet,
but every rank issues the same calls in the same order -- and the next graph begins.  `replay()` then alternates graph
replays and the recorded `fn`s.

The outcome of a recording is COLLECTIVE (as in Stage1Trainer._capture_dp): a rank whose recording throws makes up the
collectives it did not reach with dummy tensors of the recorded shapes (`expect`, taken from an eager warm-up step), then all
ranks agree on success with a MIN all-reduce, and any failure anywhere leaves every rank on eager launches.

Used by the stage-2 / stage-3 trainers (fneus/trainer2.py, trainer3.py); the stage-1 trainer has its own, older form with
the early gradient exchange on the autograd thread.
"""
from __future__ import annotations

import sys
from typing import Callable, List, Optional, Tuple

import torch


class SegmentedStep:
    def __init__(self, device):
        self.device = device
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.calls: List[Callable[[], None]] = []
        self.result = None
        self._open: Optional[torch.cuda.CUDAGraph] = None
        self._pool = None
        self.recording = False
        self.counting = False
        self.shapes: List[Tuple[tuple, torch.dtype]] = []       # (shape, dtype) of every collective of the step, in order

    # ---- called from inside the step -------------------------------------------------------------------------------
    def cut(self, fn: Callable[[], None], like: torch.Tensor):
        """a collective of the step: `fn()` reduces `like` in place"""
        if not self.recording:
            if self.counting:
                self.shapes.append((tuple(like.shape), like.dtype))
            fn()
            return
        self._open.capture_end()
        self.graphs.append(self._open)
        self._open = None
        fn()                                       # eager, every rank: keeps the ranks in lock-step while recording
        self.calls.append(fn)
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self._pool, capture_error_mode="thread_local")
        self._open = g

    # ---- recording / replay ----------------------------------------------------------------------------------------
    def count_eagerly(self, body: Callable[[], object]):
        """run `body` eagerly and note the collectives it issues (their number and shapes are fixed per step)"""
        self.shapes, self.counting = [], True
        try:
            return body()
        finally:
            self.counting = False

    def record(self, body: Callable[[], object], expect: List[Tuple[tuple, torch.dtype]], dist_active: bool) -> bool:
        """-> True when every rank recorded the step; False: every rank stays on eager launches"""
        import gc
        import torch.distributed as dist
        gc.collect()
        torch.cuda.synchronize()
        self.graphs, self.calls, self.result = [], [], None
        self._pool = torch.cuda.graph_pool_handle()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ok = False
        try:
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                # thread_local: the process group's watchdog thread may touch the runtime while this thread records
                g.capture_begin(pool=self._pool, capture_error_mode="thread_local")
                self._open, self.recording = g, True
                self.result = body()
                self.recording = False
                self._open.capture_end()
                self.graphs.append(self._open)
                self._open = None
            ok = len(self.calls) == len(expect)
            if not ok:
                print(f"[fneus] recorded {len(self.calls)} collectives, the eager step issues {len(expect)}", file=sys.stderr)
        except Exception as e:      # noqa: BLE001 -- any failure must leave a working (eager) trainer behind
            print(f"[fneus] data-parallel graph capture failed ({e!r}); continuing with eager launches", file=sys.stderr)
        finally:
            self.recording = False
            if self._open is not None:       # end the open capture ON THE STREAM THAT IS CAPTURING
                try:
                    with torch.cuda.stream(side):
                        self._open.capture_end()
                except Exception:   # noqa: BLE001
                    pass
                self._open = None
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        # make up for the collectives a failed recording did not reach, then agree on the outcome
        if dist_active:
            for shape, dtype in expect[len(self.calls):]:
                dist.all_reduce(torch.zeros(shape, dtype=dtype, device=self.device))
            flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            torch.cuda.synchronize()
            if ok and flag.item() < 0.5:
                print("[fneus] data-parallel graph capture failed on another rank; continuing with eager launches",
                      file=sys.stderr)
            ok = flag.item() >= 0.5
        if not ok:
            self.graphs, self.calls, self.result = [], [], None
        return ok

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.calls):
                self.calls[i]()
        return self.result
