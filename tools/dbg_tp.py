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
p64 = {"W": [w.double() for w in p["W"]], "b": [b.double() for b in p["b"]], "scale": 1.0}
sdf_r, feat_r, nrm_r, aux = R.sdf_value_feature_normal(x.cpu().double(), p64)
for prec in (3, 1):
    for train in (True, False):
        st = ops.SdfStash(n, dev, prec, train)
        sdf, feat, nrm = ops.sdf_fwd_grad(net.blob, n, prec, st, train, pts=x)
        torch.cuda.synchronize()
        for l in range(0, 8):
            u = aux["u"][l + 1] if l + 1 != 4 else None
            if u is None: continue
            got = st.plane(st.h[:, l]).cpu().double()[:, :u.shape[1]]
            print(f"   h_{l+1}: max err {(got - u).abs().max():.3e}")
        print(f"prec {prec} train {train}: sdf {(sdf.cpu().double()-sdf_r[:,0]).abs().max():.2e} feat {(feat.cpu().double()-feat_r).abs().max():.2e} "
              f"normal {(nrm.cpu().double()-nrm_r).abs().max():.2e}  h0 {(st.plane(st.h[:,0]).cpu().double()-aux['h'][1]).abs().max() if 'h' in aux else -1:.2e}")
