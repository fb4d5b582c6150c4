"""Round 6: phase cycles of the pipelined colour backward (FNEUS_LIB = a -DFNEUS_C8_STAMPS build), feature cotangent as fragments"""
import os, sys
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
dev = torch.device("cuda:0"); n = 65536
net = ops.PackedNet("color", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.color_state_dict(21).items()}); net.pack()
x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1).contiguous()
nrm, feat, c = torch.randn(n, 3, device=dev), (torch.randn(n, 256, device=dev) * 0.3).contiguous(), torch.randn(n, 3, device=dev)
st = ops.ColStash(n, dev, 3, gprec=2)
rgb = ops.color_fwd(net.blob, n, 3, nrm, feat, st, True, pts=x, dirs=d)
plane = torch.zeros(2 * ((n + 63) // 64), 16, 64, 8, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.color_bwd(net.blob, n, 3, c, rgb, st, dfeat_plane=plane); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.color_bwd(net.blob, n, 3, c, rgb, st, dfeat_plane=plane)
e1.record(); torch.cuda.synchronize(); print("colour backward, fragments out: %.1f us" % (e0.elapsed_time(e1) * 100))
