"""fneus_color_out_dw alone: time per launch at 65 536 samples for slice counts (FNEUS_COD_SLICES)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, synth
dev = torch.device("cuda:0")
n = 65536
net = ops.PackedNet("color", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.color_state_dict(24).items()})
st = ops.ColStash(n, dev, 3, gprec=2)
st.u.normal_(); st.u3_lo.normal_()
rgb = torch.rand(n, 3, device=dev); d_rgb = torch.randn(n, 3, device=dev)
grad = torch.zeros(net.n_params, dtype=torch.float32, device=dev)
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
for sl in (4, 8, 16, 32, 64):
    os.environ["FNEUS_COD_SLICES"] = str(sl)  # (no longer read)
    ts = []
    for rep in range(6):
        big.zero_()                       # cold caches, as inside the step (the planes were written 1 ms earlier, 5 GB of traffic ago)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.color_out_dw(net, st, d_rgb, rgb, grad, n); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"slices {sl:3d}: " + " ".join(f"{t:6.1f}" for t in ts) + " us", flush=True)
