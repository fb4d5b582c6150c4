import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
net = ops.PackedNet("sdf", dev).load_state_dict(sd); net.pack()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
prec = 3
x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
def run(train=True):
    st = ops.SdfStash(n, dev, prec, train)
    for t in (st.h, st.a, st.feat, st.pe):
        if t is not None: t.zero_()
    st.ps.zero_();
    if st.pa is not None: st.pa.zero_()
    sdf, feat, nrm = ops.sdf_fwd_grad(net.blob, n, prec, st, train, pts=x)
    torch.cuda.synchronize()
    d = {"sdf": sdf, "feat": feat, "normal": nrm, "h": st.h, "ps": st.ps}
    if train: d.update({"a": st.a, "pa": st.pa, "featp": st.feat})
    return {k: v.clone() for k, v in d.items()}
ref = run()
bad = {}
for it in range(150):
    cur = run()
    for k in ref:
        neq = (cur[k] != ref[k])
        if neq.any():
            idx = neq.nonzero()
            bad.setdefault(k, []).append((it, int(neq.sum()), idx[0].tolist(), idx[-1].tolist()))
for k, v in bad.items():
    print(k, len(v), "runs differ; first:", v[0])
print("done; tensors with differences:", list(bad.keys()))
# detail of the normal mismatches of the last differing run
cur = None
for it in range(50):
    c = run()
    if (c["normal"] != ref["normal"]).any():
        cur = c
        break
if cur is not None:
    neq = (cur["normal"] != ref["normal"]).any(dim=1).nonzero().reshape(-1).cpu().numpy()
    print("rows mod 32:", sorted(set((neq % 32).tolist())), "tiles:", sorted(set((neq // 32).tolist()))[:10], "tile mod 4:", sorted(set(((neq // 32) % 4).tolist())))
    r0 = int(neq[0])
    print("ref ", ref["normal"][r0].cpu().numpy(), "cur ", cur["normal"][r0].cpu().numpy())
    # (which of the two is right: see tests/test_hip_determinism.py / test_hip_sdf.py, which compare with the oracle)
