"""does the background network's weight-gradient record reach the SDF network's launch? (womask step, eager)"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import autograd as A, ops
from fneus.trainer import Stage1Trainer, synthetic_batches, WMASK_MODEL
dev = torch.device("cuda:0")
conf = copy.deepcopy(WMASK_MODEL); conf["neus_renderer"]["n_outside"] = 32
tr = Stage1Trainer(dev, model_conf=conf, use_graph=False)
orig_nerf, orig_run = A.NerfFn.backward, A._run_nerf_dw
def spy_run(rec):
    print("  standalone nerf launch"); return orig_run(rec)
A._run_nerf_dw = spy_run
orig_jobs = ops.sdf_dw_jobs
def spy_jobs(*a, **k):
    g = orig_jobs(*a, **k); print("  table", g.tag, len(g.jobs), "jobs", g.n_wgs, "wgs"); return g
ops.sdf_dw_jobs = spy_jobs
kw = dict(cos_anneal_ratio=0.5, background_rgb=torch.ones(1, 3, device=dev))
for i, b in enumerate(synthetic_batches(3, 512, dev)):
    print("step", i, "open", dict(A._PENDING_OPEN))
    tr.train_step(b, **kw)
    print("   after: open", dict(A._PENDING_OPEN), "pending", {k: len(v) for k, v in A._PENDING_NERF.items()})
