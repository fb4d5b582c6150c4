import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
from oracle import ref_torch as R
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
p = R.sdf_params_from_state_dict(sd)
net = ops.PackedNet("sdf", dev); net.set_raw_from_effective([w.to(dev) for w in p["W"]], [b.to(dev) for b in p["b"]]); net.pack()
n = 1000
rs = np.random.RandomState(7)
x = T(rs.uniform(-1.1, 1.1, size=(n, 3)).astype(np.float32)).to(dev)
prec = int(sys.argv[1])
st = ops.SdfStash(n, dev, prec, False)
for t in (st.h, st.pe):
    t.fill_(123.0)
sdf, feat, nrm = ops.sdf_fwd_grad(net.blob, n, prec, st, False, pts=x)
torch.cuda.synchronize()
print("sdf[:4]", sdf[:4].cpu().numpy(), "h0 row0[:4]", st.h[0, 0, 0, :4].float().cpu().numpy(), "h7", st.h[0, 7, 0, :4].float().cpu().numpy())
