"""Data-parallel semantics on ONE GPU: two half batches evaluated with the normalisers of the whole batch must add up
to exactly the loss and the parameter gradients of the whole batch (R ranks x B rays == one
R*B-ray batch).  The collectives themselves are covered by the gloo tests in tests/test_host_cpu.py."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer():
    from fneus import ops
    from fneus.trainer import Stage1Trainer, WMASK_MODEL
    conf = copy.deepcopy(WMASK_MODEL)
    conf["neus_renderer"]["perturb"] = 0.0
    return Stage1Trainer(torch.device("cuda:0"), model_conf=conf, prec=ops.PREC_PARITY, seed=5, use_graph=False)


def _grads(tr):
    return [p.grad.detach().clone() for p in tr.params if p.grad is not None]


def _clear(tr):
    for p in tr.params:
        if p.grad is not None:          # (the never-evaluated background NeRF of the wmask configuration has none)
            p.grad.zero_()


def test_two_half_batches_with_global_normalisers_equal_the_full_batch():
    from fneus.trainer import synthetic_batches
    dev = torch.device("cuda:0")
    full = synthetic_batches(1, 256, dev, seed0=777)[0]
    halves = [full[:128].contiguous(), full[128:].contiguous()]
    tr = _trainer()
    # reference: the whole batch on one "rank"
    tr._step_body(full, 1.0, None, with_optimizer=False)          # first call attaches the flat gradient buffers
    _clear(tr)
    ref_losses = tr._step_body(full, 1.0, None, with_optimizer=False)
    ref_losses = {k: float(v.detach()) for k, v in ref_losses.items()}
    ref = _grads(tr)
    _clear(tr)
    # "rank 0" and "rank 1": first collect each half's normalisers (what the all-reduce would sum) ...
    seen = []
    tr.reduce_norms = lambda n: (seen.append(n.clone()), n)[1]
    for h in halves:
        tr._step_body(h, 1.0, None, with_optimizer=False)
    _clear(tr)
    total = seen[0] + seen[1]
    assert float(total[3]) == 256.0
    # ... then run both halves against the global normalisers; the gradients accumulate like an all-reduce(SUM)
    tr.reduce_norms = lambda n: total
    parts = [tr._step_body(h, 1.0, None, with_optimizer=False) for h in halves]
    got = _grads(tr)
    for k in ("loss", "color_loss", "surface_loss", "eikonal_loss", "mask_loss"):
        s = sum(float(p[k].detach()) for p in parts)
        assert abs(s - ref_losses[k]) <= 2e-6 * max(1.0, abs(ref_losses[k])), (k, s, ref_losses[k])
    worst = 0.0
    for g, r in zip(got, ref):
        scale = r.abs().max().item() + 1e-12
        worst = max(worst, (g - r).abs().max().item() / scale)
    print(f"  two half batches vs full batch: worst relative gradient difference {worst:.2e}")
    assert worst <= 2e-3        # fp32 atomics / summation order only (the K2-K3 chains are evaluated per sample)


def test_config5_shape_four_shards_of_512_womask_rays_equal_the_2048_ray_batch():
    """BASELINE config 5's batch (womask.conf: 64 + 64 + 32 outside samples, background NeRF++, no mask loss; 2048 rays) as
    four ray shards of 512: with the global loss normalisers the shards' losses and gradients -- incl. the background
    network's -- add up to those of the one 2048-ray batch"""
    from fneus import ops
    from fneus.trainer import Stage1Trainer, WMASK_MODEL, synthetic_batches
    dev = torch.device("cuda:0")
    conf = copy.deepcopy(WMASK_MODEL)
    conf["neus_renderer"].update(n_outside=32, perturb=0.0)
    tr = Stage1Trainer(dev, model_conf=conf, prec=ops.PREC_PARITY, seed=6, use_graph=False, mask_weight=0.0)
    full = synthetic_batches(1, 2048, dev, seed0=4242)[0]
    shards = [full[i * 512:(i + 1) * 512].contiguous() for i in range(4)]
    bg = torch.ones(1, 3, device=dev)
    tr._step_body(full, 0.5, bg, with_optimizer=False)
    _clear(tr)
    ref_losses = {k: float(v.detach()) for k, v in tr._step_body(full, 0.5, bg, with_optimizer=False).items()}
    ref = _grads(tr)
    assert len(ref) == len(tr.params)                      # the background network takes part
    _clear(tr)
    seen = []
    tr.reduce_norms = lambda n: (seen.append(n.clone()), n)[1]
    for h in shards:
        tr._step_body(h, 0.5, bg, with_optimizer=False)
    _clear(tr)
    total = sum(seen[1:], seen[0])
    assert float(total[3]) == 2048.0
    tr.reduce_norms = lambda n: total
    parts = [tr._step_body(h, 0.5, bg, with_optimizer=False) for h in shards]
    got = _grads(tr)
    for k in ("loss", "color_loss", "surface_loss", "eikonal_loss"):
        s = sum(float(p[k].detach()) for p in parts)
        assert abs(s - ref_losses[k]) <= 4e-6 * max(1.0, abs(ref_losses[k])), (k, s, ref_losses[k])
    worst = max((g - r).abs().max().item() / (r.abs().max().item() + 1e-12) for g, r in zip(got, ref))
    print(f"  config-5 shape, four shards vs the 2048-ray batch: worst relative gradient difference {worst:.2e}")
    assert worst <= 2e-3


def _two_ranks(graph: bool, fail_rank: int = -1, stage: int = 1, conf: str = "wmask", steps: int = 7):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + (os.getpid() + 151 * int(graph) + 53 * stage + 17 * (fail_rank + 1)) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "dp_check.py")]
    env = dict(os.environ, DP_CHECK_GRAPH="1" if graph else "0", DP_CHECK_FAIL_RANK=str(fail_rank), DP_CHECK_STAGE=str(stage),
               DP_CHECK_CONF=conf, DP_CHECK_STEPS=str(steps))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=360, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("DP_CHECK")]
    assert r.returncode == 0 and line and line[0].endswith("OK"), (r.stdout[-2000:], r.stderr[-2000:])
    print(" ", line[0])
    return [float(v) for v in [l for l in r.stdout.splitlines() if l.startswith("DP_TRACE")][0].split()[1:]]


def test_two_ranks_sharing_the_gpu_stay_identical():
    """the real N > 1 code path (broadcast, loss-normaliser all-reduce, arena all-reduce, FlatAdam) with two processes;
    gloo instead of RCCL because both ranks have to share the one GPU of the test box.  Eager launches, then the same run
    with the step replayed as three hipGraphs around the two collectives: the global loss trajectories must agree."""
    eager = _two_ranks(graph=False)
    graphed = _two_ranks(graph=True)
    assert len(eager) == len(graphed) == 7
    worst = max(abs(a - b) / max(abs(a), 1e-2) for a, b in zip(eager, graphed))
    print(f"  eager vs three-graph global loss trajectories: worst relative difference {worst:.2e}")
    # Two runs differ by the order of the fp32 atomics in the weight-gradient GEMM; Adam and the ill-conditioned sampler
    # amplify that over the steps (observed over many runs: 1e-6 ... 2e-4, rare outliers beyond): the first steps must agree
    # closely, the later ones must stay on the same curve
    for i, (a, b) in enumerate(zip(eager, graphed)):
        assert abs(a - b) <= (1e-3 if i < 3 else 3e-2) * max(abs(a), 1e-2), (i, a, b)


def test_exchange_form_is_measured_and_agreed_at_start_up():
    """more than one rank, FNEUS_DP_EARLY unset: the job times both forms of its gradient exchange on its own training steps (the
    arena in two parts beside the SDF backward: four graphs around three collectives; one exchange behind the backward: three
    around two), every rank sees the same two MAX-over-ranks timings and keeps the same form (tools/dp_check.py checks the
    agreement, that both timings exist and that the replicas are bit-identical across the switch); 60 steps of two ranks."""
    trace = _two_ranks(graph=True, steps=60)
    assert len(trace) == 60 and all(t == t for t in trace)


def test_capture_failure_on_one_rank_moves_every_rank_to_eager_launches():
    """the graph capture of rank 1 throws after the first exchange: the outcome must be collective (every rank issues the
    same three collectives during capture and all fall back to eager launches), the replicas stay bit-identical and the
    run finishes -- instead of rank 1's first real collectives pairing with rank 0's dummy ones"""
    trace = _two_ranks(graph=True, fail_rank=1)
    eager = _two_ranks(graph=False)
    assert len(trace) == 7
    assert abs(trace[0] - eager[0]) <= 1e-5 * max(1.0, abs(eager[0]))


class _FakeReduce:
    """stands in for the all-reduce(SUM) of the data-parallel trainers on one process: pass 1 records what every "rank"
    contributes (per call site, in call order), pass 2 hands back the sums"""

    def __init__(self):
        self.rec, self.tot, self.i, self.apply = [], None, 0, False

    def begin_rank(self):
        self.i = 0

    def __call__(self, t):
        if not self.apply:
            if self.i == len(self.rec):
                self.rec.append(t.detach().clone())
            else:
                self.rec[self.i] += t.detach()
            self.i += 1
            return t
        out = self.rec[self.i].clone()
        self.i += 1
        return out


def _param_grads(params):
    return [p.grad.detach().clone() for p in params]


def test_stage2_two_half_batches_with_global_hit_count_equal_the_full_batch():
    """stage 2 data parallel (fneus/trainer2.py): R ranks x B rays == one R*B-ray batch"""
    from fneus.trainer import synthetic_batches
    from fneus.trainer2 import Stage2Trainer, stage2_loss
    dev = torch.device("cuda:0")
    tr = Stage2Trainer(dev, seed=4)
    full = synthetic_batches(1, 128, dev, seed0=321)[0]
    g = torch.Generator().manual_seed(1)
    ut, uz = torch.rand(128, 4, generator=g).to(dev), torch.rand(128, 4, generator=g).to(dev)

    def run(rows, reduce):
        d = full[rows].contiguous()
        out = tr.renderer.lvis_render(d[:, :3].contiguous(), d[:, 3:6].contiguous(), None, None, u_theta=ut[rows].contiguous(),
                                      u_z=uz[rows].contiguous(), fixed_shape=True)
        return stage2_loss(out, reduce)

    for p in tr.params:
        p.grad = None
    L = run(slice(0, 128), None)
    L["loss"].backward()
    ref, ref_loss = _param_grads(tr.params), float(L["loss"])
    fake = _FakeReduce()
    for rows in (slice(0, 64), slice(64, 128)):
        fake.begin_rank()
        run(rows, fake)
    fake.apply = True
    for p in tr.params:
        p.grad = None
    total = 0.0
    for rows in (slice(0, 64), slice(64, 128)):
        fake.begin_rank()
        L = run(rows, fake)
        L["loss"].backward()                      # .grad accumulates like the arena all-reduce(SUM)
        total += float(L["loss"])
    assert abs(total - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss)), (total, ref_loss)
    worst = max(((a - b).abs().max() / (b.abs().max() + 1e-12)).item() for a, b in zip(_param_grads(tr.params), ref))
    print(f"  stage 2: two half batches vs full batch: worst relative gradient difference {worst:.2e}")
    assert worst <= 1e-4
    # the trainer's own data-parallel step (one rank: the collectives are identities) runs and keeps .grad in its arena
    trd = Stage2Trainer(dev, seed=4, distributed=True)
    o = trd.train_step(full)
    assert bool(torch.isfinite(o["loss"])) and trd.iter_step == 1
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(trd.grads.small, trd.grads._small_views))


def test_stage3_two_half_batches_with_global_statistics_equal_the_full_batch():
    """stage 3 data parallel (fneus/trainer3.py): mask sum and the latent-sparsity statistics are global"""
    from fneus.trainer import synthetic_batches
    from fneus.trainer3 import Stage3Trainer, stage3_loss
    dev = torch.device("cuda:0")
    tr = Stage3Trainer(dev, seed=4)
    full = synthetic_batches(1, 128, dev, seed0=322)[0]
    g = torch.Generator().manual_seed(2)
    ut, up = torch.rand(128, 32, generator=g).to(dev), torch.rand(128, 32, generator=g).to(dev)

    def run(rows, reduce):
        d = full[rows].contiguous()
        tr.mateIllu_network.stat_reduce = reduce
        out = tr.renderer.mateIllu_render(d[:, :3].contiguous(), d[:, 3:6].contiguous(), None, None, u_theta=ut, u_phi=up,
                                          fixed_shape=True)
        return stage3_loss(out, d[:, 6:9].contiguous(), (d[:, 9:10] > 0.5).float(), reduce)

    for p in tr.params:
        p.grad = None
    L = run(slice(0, 128), None)
    L["loss"].backward()
    ref, ref_rgb, ref_kl = _param_grads(tr.params), float(L["rgb_loss"]), float(L["encoder_loss"])
    fake = _FakeReduce()
    for rows in (slice(0, 64), slice(64, 128)):
        fake.begin_rank()
        run(rows, fake)
    fake.apply = True
    for p in tr.params:
        p.grad = None
    rgb = 0.0
    for rows in (slice(0, 64), slice(64, 128)):
        fake.begin_rank()
        L = run(rows, fake)
        L["loss"].backward()
        rgb += float(L["rgb_loss"])
        assert abs(float(L["encoder_loss"]) - ref_kl) <= 1e-6 * max(1.0, abs(ref_kl))       # the same global value on every rank
    assert abs(rgb - ref_rgb) <= 2e-6 * max(1.0, abs(ref_rgb))
    worst = max(((a - b).abs().max() / (b.abs().max() + 1e-12)).item() for a, b in zip(_param_grads(tr.params), ref))
    print(f"  stage 3: two half batches vs full batch: worst relative gradient difference {worst:.2e}")
    assert worst <= 1e-4
    trd = Stage3Trainer(dev, seed=4, distributed=True)
    o = trd.train_step(full)
    assert bool(torch.isfinite(o["loss"])) and trd.iter_step == 1


@pytest.mark.parametrize("stage", [1, 2, 3])
def test_one_rank_over_rccl_runs_the_data_parallel_call_sequence(stage):
    """RCCL refuses two ranks on one device, so the test box cannot hold a real N > 1 job; FNEUS_DP_SINGLE=1 makes a world of
    ONE rank issue every collective of the data-parallel step anyway: communicator set-up, the parameter broadcast, the
    loss-normaliser all-reduce, the async all-reduce of the early arena part on the side stream while the SDF backward
    runs, the late part -- eagerly and between the four hipGraph segments.  The trajectories must equal each other (a sum
    over one rank changes nothing) and the graphs must have been captured."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traces = []
    for graph in (False, True):
        for attempt in range(2):     # (one transient failure of the child's communicator set-up was seen in ~10 runs of the suite:
            #                           a second attempt on another port before the test gives up)
            port = 29900 + (os.getpid() + 37 * int(graph) + 11 * stage + 45 * attempt) % 90
            env = dict(os.environ, DP_CHECK_GRAPH="1" if graph else "0", DP_CHECK_BACKEND="nccl", FNEUS_DP_SINGLE="1",
                       DP_CHECK_STAGE=str(stage),
                       RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            try:
                r = subprocess.run([sys.executable, os.path.join(root, "tools", "dp_check.py")], capture_output=True, text=True,
                                   timeout=300, env=env)
            except subprocess.TimeoutExpired:
                if attempt == 0:
                    continue
                raise
            line = [l for l in r.stdout.splitlines() if l.startswith("DP_CHECK")]
            if r.returncode == 0 and line and line[0].endswith("OK"):
                break
            if attempt == 0:
                print("  first attempt failed:", r.stdout[-500:], r.stderr[-500:])
        assert r.returncode == 0 and line and line[0].endswith("OK"), (r.stdout[-2000:], r.stderr[-2000:])
        print(" ", line[0])
        traces.append([float(v) for v in [l for l in r.stdout.splitlines() if l.startswith("DP_TRACE")][0].split()[1:]])
    if stage == 1:       # (stages 2 / 3 draw their directions differently in eager and recorded steps)
        for i, (a, b) in enumerate(zip(*traces)):
            assert abs(a - b) <= (1e-3 if i < 3 else 3e-2) * max(abs(a), 1e-2), (i, a, b)


@pytest.mark.parametrize("stage", [2, 3])
def test_stages_2_and_3_two_ranks_with_graph_chains_stay_identical(stage):
    """the stage-2 / stage-3 data-parallel step as a chain of hipGraphs cut at its collectives (fneus/seggraph.py: the loss
    normalisers -- stage 3 also the latent-sparsity statistics -- and the gradient buffer), two processes sharing the GPU over
    gloo: the replicas stay bit-identical, 3 / 4 graphs are recorded; then with a recording failure injected on rank 1: every
    rank falls back to eager launches together and the run finishes with identical replicas"""
    trace = _two_ranks(graph=True, stage=stage)
    assert len(trace) == 7 and all(np.isfinite(trace))
    _two_ranks(graph=True, fail_rank=1, stage=stage)
    _two_ranks(graph=False, stage=stage)


def test_two_ranks_womask_configuration_stay_identical():
    """the womask configuration data parallel: the background NeRF's gradients are part of the arena (late part, with the SDF's)"""
    eager = _two_ranks(graph=False, conf="womask")
    graphed = _two_ranks(graph=True, conf="womask")
    for i, (a, b) in enumerate(zip(eager, graphed)):
        assert abs(a - b) <= (1e-3 if i < 3 else 3e-2) * max(abs(a), 1e-2), (i, a, b)
