#!/usr/bin/env python3
"""K1 two-pass pipelined kernel: time at n points (timing variants through FNEUS_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, synth
dev = torch.device("cuda:0")
os.environ["FNEUS_K1_W8_BIG"] = os.environ.get("FNEUS_K1_W8_BIG", "3")
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()
for n in [int(a) for a in sys.argv[1:]] or [65536]:
    x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
    for _ in range(5): ops.sdf_fwd(net.blob, n, 3, pts=x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.sdf_fwd(net.blob, n, 3, pts=x)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(os.environ.get('FNEUS_LIB', 'default')):40s} n={n}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)
