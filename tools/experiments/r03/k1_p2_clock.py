#!/usr/bin/env python3
"""In-kernel clock of the two-pass K1 (variants built with -DFNEUS_P2_CLOCK): shader cycles / 100 MHz ticks per wave."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
dev = torch.device("cuda:0")
os.environ["FNEUS_K1_W8_BIG"] = "3"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()
x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
n4 = (n + 3) & ~3
out = torch.zeros(n4 + 256 * 4 * 2 * 2, dtype=torch.float32, device=dev)
for _ in range(50): ops.sdf_fwd(net.blob, n, 3, pts=x, out=out)          # warm: let the clock settle under load
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): ops.sdf_fwd(net.blob, n, 3, pts=x, out=out)
e1.record(); torch.cuda.synchronize()
st = out[n4:].cpu().numpy().view(np.uint64).reshape(256, 4, 2).astype(np.float64)
cyc, ticks = st[..., 0].mean(), st[..., 1].mean()
print(f"{os.path.basename(os.environ.get('FNEUS_LIB', 'default')):36s} n={n}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us per launch; per wave {cyc:9.0f} cycles in {ticks / 100:7.1f} us -> {cyc / ticks * 100:6.0f} MHz")
