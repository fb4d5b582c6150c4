#!/usr/bin/env python3
"""Phase budget of the staggered-halves K1 (s_memtime stamps; build_variant.sh stamps "-DFNEUS_W8_STAMPS" sdf_w8_kernels.hip).
Usage: k1_s8_stamps.py n"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
dev = torch.device("cuda:0")
n = int(sys.argv[1])
os.environ["FNEUS_K1_W8_BIG"] = "22"
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()
x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
n4 = (n + 3) & ~3
out = torch.zeros(n4 + 32 * 8 * 8 * 5 * 2, dtype=torch.float32, device=dev)
for _ in range(3):
    ops.sdf_fwd(net.blob, n, 3, pts=x, out=out)
torch.cuda.synchronize()
st = out[n4:].cpu().numpy().view(np.uint64).reshape(32, 8, 8, 5).astype(np.int64)
print(f"s8 HB=2 n={n}: cycles, mean over 32 workgroups; group 0 = waves 0-3, group 1 = waves 4-7")
for g in (0, 1):
    s = st[:, 4 * g:4 * g + 4]
    print(f" group {g}")
    for l in range(8):
        D = (s[:, :, l, 1] - s[:, :, l, 0]).mean(); b1 = (s[:, :, l, 2] - s[:, :, l, 1]).mean()
        P = (s[:, :, l, 3] - s[:, :, l, 2]).mean(); b2 = (s[:, :, l, 4] - s[:, :, l, 3]).mean()
        print(f"  layer {l}: D {D:7.0f}  wait {b1:7.0f}  P {P:7.0f}  wait {b2:7.0f}   | layer {(s[:, :, l, 4] - s[:, :, l, 0]).mean():7.0f}")
t0 = st[:, :, 0, 0].min(axis=1); t1 = st[:, :, 7, 4].max(axis=1)
print(f" first -> last stamp of a workgroup {float((t1 - t0).mean()):9.0f} cycles; MFMA issue alone {(3 + 16 * 6 + 17) * 12 * 32 * 2}")
# timeline of workgroup 0, waves 0 and 4, relative
b = 0
base = st[b, :, 0, 0].min()
for wv in (0, 4):
    print(f" wg 0 wave {wv}: " + " | ".join(f"L{l} D {st[b, wv, l, 0] - base}-{st[b, wv, l, 1] - base} P {st[b, wv, l, 2] - base}-{st[b, wv, l, 3] - base}" for l in range(8)))
