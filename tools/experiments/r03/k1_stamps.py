#!/usr/bin/env python3
"""Per-phase cycle budget of K1 on 8-wave workgroups, from s_memtime stamps (build: tools/experiments/build_variant.sh stamps
"-DFNEUS_W8_STAMPS" sdf_w8_kernels.hip; run with FNEUS_LIB=.../libfneus_stamps.so).  Usage: k1_stamps.py HB n"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
dev = torch.device("cuda:0")
hb, n = int(sys.argv[1]), int(sys.argv[2])
os.environ["FNEUS_K1_W8_BIG"] = str(hb) if hb > 1 else "0"
os.environ["FNEUS_K1_W8_SMALL"] = os.environ.get("W8_SMALL_MODE", "1") if hb == 1 else "0"
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()
x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
n4 = (n + 3) & ~3
extra = 32 * 8 * 8 * 5 * 2                       # uint64 stamps as pairs of floats
out = torch.zeros(n4 + extra, dtype=torch.float32, device=dev)
for _ in range(3):
    ops.sdf_fwd(net.blob, n, 3, pts=x, out=out)
torch.cuda.synchronize()
st = out[n4:].cpu().numpy().view(np.uint64).reshape(32, 8, 8, 5).astype(np.int64)
nb = min(32, (n + 32 * hb - 1) // (32 * hb))
st = st[:nb]
names = ["dense (bias + weights + MFMAs)", "softplus", "wait at barrier 1", "fragments -> LDS + barrier 2"]
print(f"HB={hb} n={n}: cycles per layer (mean over {nb} workgroups x 8 waves; wave 7 of layer 3 moves the skip input)")
tot = np.zeros(4)
for l in range(8):
    d = [st[:, :, l, 1] - st[:, :, l, 0], st[:, :, l, 2] - st[:, :, l, 1]]
    if l < 7:
        d += [st[:, :7, l, 3] - st[:, :7, l, 2], st[:, :7, l, 4] - st[:, :7, l, 3]]
    else:
        d += [np.zeros(1), np.zeros(1)]
    m = [float(a.mean()) for a in d]
    tot += np.array(m)
    print(f"  layer {l}: " + "  ".join(f"{nm.split(' ')[0]} {v:8.0f}" for nm, v in zip(names, m)) + f"   | whole layer {float((st[:, :, l, 4 if l < 7 else 2] - st[:, :, l, 0]).mean()):8.0f}")
print("  sum:     " + "  ".join(f"{nm} {v:8.0f}" for nm, v in zip(names, tot)))
print(f"  first stamp -> last stamp of a workgroup: {float((st[:, :, 7, 2].max(axis=1) - st[:, :, 0, 0].min(axis=1)).mean()):9.0f} cycles "
      f"(MFMA issue alone: {(3 + 16 * 6 + 17) * 3 * hb * 32 * 2} per SIMD with two waves)")
