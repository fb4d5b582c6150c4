import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(21).items()}); net.pack()
n = 40067
x = (torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 2 - 1).contiguous()
os.environ["FNEUS_R8_NH"] = "4"
for r8 in (0, 1):
    os.environ["FNEUS_K2_REV8"] = str(r8)
    st = ops.SdfStash(n, dev, 3, True, 1)
    st.a.zero_()
    ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
    torch.cuda.synchronize()
    a = st.a[0].float()      # [8, T, 16, 64, 8]
    print("r8", r8, "tiles with samples", st.tiles, "allocated", a.shape[1])
    for t in range(st.tiles - 2, a.shape[1]):
        nz = (a[:, t].abs() > 0).reshape(8, -1).sum(1).tolist()
        print("  tile", t, "nonzeros per slot", nz)
    print("  tile 0:", (a[:, 0].abs() > 0).reshape(8, -1).sum(1).tolist(), " tile 1:", (a[:, 1].abs() > 0).reshape(8, -1).sum(1).tolist())
