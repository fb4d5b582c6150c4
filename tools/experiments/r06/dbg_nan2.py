import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops
from fneus.trainer import Stage1Trainer, synthetic_batches
dev = torch.device("cuda:0")
batches = synthetic_batches(5, 256, dev, seed0=77)
def run(rider, planes):
    ops.DEFAULT_FOLD_RIDER = rider
    ops.FEAT_PLANES = planes
    torch.manual_seed(11)
    tr = Stage1Trainer(dev, seed=5, use_graph=False)
    return [float(tr.train_step(b)["loss"].detach()) for b in batches]
for seq in sys.argv[1:]:
    rider, planes = seq[0] == "1", seq[1] == "1"
    print(seq, run(rider, planes), flush=True)
