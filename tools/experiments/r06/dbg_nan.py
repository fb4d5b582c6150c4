import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops
from fneus.trainer import Stage1Trainer, synthetic_batches
dev = torch.device("cuda:0")
for planes in (False, True):
    ops.FEAT_PLANES = planes
    torch.manual_seed(11)
    tr = Stage1Trainer(dev, seed=5, use_graph=(len(sys.argv) > 1))
    batches = synthetic_batches(6, 256, dev, seed0=77)
    for i, b in enumerate(batches):
        out = tr.train_step(b)
        bad = [n for m in tr.modules for n, p in m.named_parameters() if not torch.isfinite(p).all()]
        print(f"planes {planes} step {i} loss {float(out['loss']):.6f} variance {float(tr.deviation_network.variance):.6f} non-finite params: {bad[:4]}", flush=True)
    st = tr.sdf_network._ws.cache[("sdf_stash", 256 * 128, 3, True)]
    print("   feat planes finite:", bool(torch.isfinite(st.feat.float()).all()), "shape", tuple(st.feat.shape))
print("---- rider off")
ops.FEAT_PLANES = True
ops.DEFAULT_FOLD_RIDER = False
torch.manual_seed(11)
tr = Stage1Trainer(dev, seed=5, use_graph=False)
for i, b in enumerate(synthetic_batches(5, 256, dev, seed0=77)):
    out = tr.train_step(b)
    bad = [n for m in tr.modules for n, p in m.named_parameters() if not torch.isfinite(p).all()]
    print(f"rider off step {i} loss {float(out['loss'].detach()):.6f} variance {float(tr.deviation_network.variance.detach()):.6f} grad {tr.deviation_network.variance.grad} non-finite: {bad[:4]}", flush=True)
