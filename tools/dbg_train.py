import sys, os, copy, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import torch
from fneus import ops, synth
from fneus.trainer import Stage1Trainer, WMASK_MODEL
dev = torch.device("cuda:0")
conf = copy.deepcopy(WMASK_MODEL)
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 16
conf["neus_renderer"] = dict(n_samples=ns, n_importance=ns, n_outside=0, up_sample_steps=4, perturb=float(os.environ.get("PERTURB","0")))
use_graph = sys.argv[1] == "1"
B = int(sys.argv[2])
tr = Stage1Trainer(dev, model_conf=conf, prec=ops.PREC_PARITY, seed=30, lr=5e-4, use_graph=use_graph)
for i in range(6):
    b = torch.from_numpy(synth.ray_batch(B, seed=900 + i, n_miss=int(os.environ.get("NMISS","3")))).to(dev)
    out = tr.train_step(b)
    torch.cuda.synchronize()
    print(i, float(out["loss"].detach()), flush=True)
