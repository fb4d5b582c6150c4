"""bitwise reproducibility of K1 (fneus_sdf_fwd) over repeated launches, small (tensor-parallel) and large launches"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
T = lambda a: torch.from_numpy(np.asarray(a))
dev = torch.device("cuda:0")
sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
net = ops.PackedNet("sdf", dev).load_state_dict(sd); net.pack()
for n in (8192, 4000, 65536):
    x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
    for prec in (3, 1):
        ref = ops.sdf_fwd(net.blob, n, prec, pts=x).clone()
        bad = 0
        for it in range(300):
            cur = ops.sdf_fwd(net.blob, n, prec, pts=x)
            if not torch.equal(cur, ref):
                bad += 1
        print(f"K1 n={n} prec={prec}: {bad} of 300 launches differ")
