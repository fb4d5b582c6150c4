for v in t2first t2first_not1; do echo "== $v"; FNEUS_LIB=$PWD/factored-neus_amd/fneus/variants/libfneus_$v.so python3 tools/experiments/r05/k1_h6_time.py 2>&1 | grep "n 65536" | head -1; done
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
from oracle import ref_torch as R
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}
p = R.sdf_params_from_state_dict(sd)
net = ops.PackedNet("sdf", dev)
net.set_raw_from_effective([w.to(dev) for w in p["W"]], [b.to(dev) for b in p["b"]])
net.pack()
for n in (128,):
    x = (torch.rand(n, 3, device=dev) * 2.2 - 1.1).contiguous()
    o = ops.sdf_fwd_h6(net.blob, n, pts=x)
    ref = ops.sdf_fwd(net.blob, n, 3, pts=x)
    bad = ~torch.isfinite(o)
    print(n, "nan count", int(bad.sum()), "first bad", bad.nonzero()[:8].flatten().tolist(), "max err finite", float((o - ref)[~bad].abs().max()) if (~bad).any() else None)
    if n == 65536:
        idx = bad.nonzero().flatten()
        print("bad per unit of 128:", torch.bincount(idx // 128, minlength=512).nonzero().flatten()[:20].tolist(), "bad tiles within unit", torch.bincount((idx % 128) // 32, minlength=4).tolist())
PY
