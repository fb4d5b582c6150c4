#!/usr/bin/env python3
"""Two (or more) data-parallel ranks of the stage-1 trainer, all on GPU 0 over gloo (RCCL refuses several ranks per
device): a functional check of the N > 1 code path -- the replicas must stay bit-identical.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
import torch.distributed as dist

from fneus.parallel import broadcast_parameters, init_from_env
from fneus.trainer import Stage1Trainer, synthetic_batches, WMASK_MODEL

# DP_CHECK_BACKEND=nccl with FNEUS_DP_SINGLE=1 and ONE rank: the same call sequence over RCCL (tests/test_hip_dp.py)
rank, world, _ = init_from_env(os.environ.get("DP_CHECK_BACKEND", "gloo"))
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
use_graph = os.environ.get("DP_CHECK_GRAPH", "0") == "1"      # four hipGraphs per step around the three collectives
stage = int(os.environ.get("DP_CHECK_STAGE", "1"))
if stage in (2, 3):
    # stages 2 / 3 (fneus/trainer2.py, trainer3.py): the fixed-shape step, eagerly or as a chain of graphs cut at the
    # collectives (fneus/seggraph.py); DP_CHECK_FAIL_RANK: that rank's recording throws in the optimiser step (after every exchange)
    from fneus.trainer2 import Stage2Trainer
    from fneus.trainer3 import Stage3Trainer
    tr = (Stage2Trainer if stage == 2 else Stage3Trainer)(dev, seed=rank, distributed=True, use_graph=use_graph)
    mods = [tr.lvis_network, tr.indiLgt_network] if stage == 2 else [tr.mateIllu_network]
    broadcast_parameters(mods)
    fail_rank = int(os.environ.get("DP_CHECK_FAIL_RANK", "-1"))
    if use_graph and rank == fail_rank:
        real = tr.optimizer.step

        def failing(*a, **k):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("injected capture failure")
            return real(*a, **k)

        tr.optimizer.step = failing
    torch.manual_seed(100 + rank)
    trace = []
    for b in synthetic_batches(7, 128, dev, rank=rank):
        out = tr.train_step(b)
        v = out["loss"].detach().clone().reshape(1)
        dist.all_reduce(v)
        trace.append(float(v))
    flat = torch.cat([p.detach().reshape(-1) for p in tr.params])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    worst = max((g - gathered[0]).abs().max().item() for g in gathered)
    n_graphs = len(tr._seg.graphs)
    want = 0 if (not use_graph or fail_rank >= 0) else (3 if stage == 2 else 4)
    flags = [None] * world
    dist.all_gather_object(flags, bool(n_graphs == want and (tr.use_graph == (want > 0))))
    ok = worst == 0.0 and all(map(lambda x: x == x and abs(x) < 1e6, trace)) and all(flags)
    if rank == 0:
        print("DP_TRACE " + " ".join(f"{v:.6f}" for v in trace))
        print(f"DP_CHECK stage={stage} world={world} graphs={n_graphs} max replica difference {worst:.3e} global loss {trace[-1]:.6f} "
              f"{'OK' if ok else 'FAILED'}")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)
import copy
conf = copy.deepcopy(WMASK_MODEL)
conf["neus_renderer"]["perturb"] = 0.0                         # no depth jitter: eager and replayed runs draw it differently
womask = os.environ.get("DP_CHECK_CONF", "wmask") == "womask"  # + 32 outside samples: the background NeRF's gradients join the arena
if womask:
    conf["neus_renderer"]["n_outside"] = 32
tr = Stage1Trainer(dev, model_conf=conf, seed=rank, distributed=True, use_graph=use_graph,
                   mask_weight=0.0 if womask else 0.1)          # different initial weights per rank on purpose ...
broadcast_parameters(tr.modules)                               # ... rank 0's must win
# DP_CHECK_FAIL_RANK=r: the graph capture of rank r throws half way (after the first exchange): every rank must fall back to
# eager launches together and the replicas must stay identical
fail_rank = int(os.environ.get("DP_CHECK_FAIL_RANK", "-1"))
if use_graph and rank == fail_rank:
    real_step = tr.optimizer.step

    def failing_step(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("injected capture failure")
        return real_step(*a, **k)

    tr.optimizer.step = failing_step
losses = None
trace = []
n_steps = int(os.environ.get("DP_CHECK_STEPS", "7"))          # >= 55: the run crosses the start-up selection of the exchange form
bs = synthetic_batches(min(n_steps, 8), 128, dev, rank=rank)
for i in range(n_steps):
    losses = tr.global_losses(tr.train_step(bs[i % len(bs)]))
    trace.append(float(losses["loss"]))
flat = torch.cat([p.detach().reshape(-1) for p in tr.params])
gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
worst = max((g - gathered[0]).abs().max().item() for g in gathered)
if fail_rank >= 0:
    graphs_ok = not tr.use_graph                      # the injected failure must have switched EVERY rank to eager launches
else:
    # one captured step per exchange form met so far (FNEUS_DP_EARLY unset: both forms are measured at start-up, fneus/trainer.py)
    graphs_ok = not use_graph or (tr.use_graph and len(tr._graphs) in (1, 2))
if n_steps >= 55 and os.environ.get("FNEUS_DP_EARLY", "auto") not in ("0", "1"):
    # the selection has finished, with timings for both forms, and every rank has taken the same decision
    c = tr.exchange_choice
    picks = [None] * world
    dist.all_gather_object(picks, None if c is None else (c["choice"], c["ms_split"], c["ms_single"]))
    graphs_ok = graphs_ok and c is not None and c["steps"] > 0 and all(p == picks[0] for p in picks) and tr._auto is None
    if rank == 0:
        print(f"DP_EXCHANGE {c}")
flags = [None] * world
dist.all_gather_object(flags, bool(graphs_ok))
ok = worst == 0.0 and bool(torch.isfinite(losses["loss"])) and all(flags)
if rank == 0:
    print("DP_TRACE " + " ".join(f"{v:.6f}" for v in trace))
    print(f"DP_CHECK world={world} graphs={int(use_graph)} max replica difference {worst:.3e} global loss {float(losses['loss']):.6f} "
          f"{'OK' if ok else 'FAILED'}")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
