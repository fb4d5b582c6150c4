"""decode the h6 blob on the host and compare it with the weights it was packed from (fp16 hi + fp6 Q(W), Q(Wl) with their scales)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth, netdesc
from oracle import ref_torch as R
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}
p = R.sdf_params_from_state_dict(sd)
net = ops.PackedNet("sdf", dev)
net.set_raw_from_effective([w.to(dev) for w in p["W"]], [b.to(dev) for b in p["b"]])
net.pack()
hb = ops.h6_blob(net.blob).cpu().numpy()
blob = net.blob.cpu().numpy()
ly = netdesc.query_layout(0)
def fp6(c):
    s, e, m = c >> 5, (c >> 3) & 3, c & 7
    v = np.where(e == 0, m / 8.0, (1 + m / 8.0) * 2.0 ** (e - 1.0))
    return np.where(s == 1, -v, v)
def unpack6(words):            # [64, 6] uint32 -> [64, 32] codes
    out = np.zeros((64, 32), dtype=np.int64)
    w = words.astype(np.uint64)
    for j in range(32):
        bit = 6 * j
        lo = w[:, bit >> 5] >> np.uint64(bit & 31)
        if (bit & 31) > 26:
            lo = lo | (w[:, (bit >> 5) + 1] << np.uint64(32 - (bit & 31)))
        out[:, j] = (lo & np.uint64(63)).astype(np.int64)
    return out
# h6 layout (h6_engine.h)
geom = [(3, 8), (16, 8), (16, 8), (16, 7), (17, 8), (16, 8), (16, 8), (16, 8)]
nblk = [1, 4, 4, 4, 5, 4, 4, 4]
off = 0
L = []
for l in range(8):
    hi = off; off += geom[l][0] * geom[l][1] * 1024
    rec = off; off += nblk[l] * geom[l][1] * 3328
    L.append((hi, rec))
l, b, t = 1, 2, 3
ks_list = [4 * b + s for s in range(4)]
nt = geom[l][1]
fwd_hi, fwd_lo = int(ly.off[l][0]), int(ly.off[l][1])
W = np.zeros((64, 32), dtype=np.float64)
for s, ks in enumerate(ks_list):
    o = (ks * nt + t) * 1024
    h = torch.from_numpy(blob[fwd_hi + o: fwd_hi + o + 1024].copy()).view(torch.bfloat16).float().numpy().reshape(64, 8)
    lo = torch.from_numpy(blob[fwd_lo + o: fwd_lo + o + 1024].copy()).view(torch.bfloat16).float().numpy().reshape(64, 8)
    W[:, 8 * s: 8 * s + 8] = (h + lo)
Wh = W.astype(np.float16).astype(np.float64)
Wl = W - Wh
rec = hb[L[l][1] + (b * nt + t) * 3328: L[l][1] + (b * nt + t + 1) * 3328]
wa = rec[0:1024].view(np.uint32).reshape(64, 4); wb = rec[1024:1536].view(np.uint32).reshape(64, 2)
la = rec[1536:2560].view(np.uint32).reshape(64, 4); lb = rec[2560:3072].view(np.uint32).reshape(64, 2)
sc = rec[3072:3328].view(np.uint32)
order = np.array([(jj >> 1) + 16 * (jj & 1) for jj in range(32)])      # element jj' = 2 e + i holds linear index 16 i + e
q = fp6(unpack6(np.concatenate([wa, wb], 1))) * (2.0 ** ((sc & 255).astype(np.float64) - 127))[:, None]
ql = fp6(unpack6(np.concatenate([la, lb], 1))) * (2.0 ** (((sc >> 8) & 255).astype(np.float64) - 127))[:, None]
print("hi16 frags: max |f16(W) - stored|", max(np.abs(hb[L[l][0] + (ks * nt + t) * 1024: L[l][0] + (ks * nt + t + 1) * 1024].view(np.float16).reshape(64, 8).astype(np.float64) - Wh[:, 8 * s: 8 * s + 8]).max() for s, ks in enumerate(ks_list)))
print("Q(W) : max |decoded - W| / max|W| per lane:", (np.abs(q - W[:, order]).max(1) / np.abs(W).max(1)).max(), " (fp6: <= ~0.07)")
print("Q(Wl): max |decoded - Wl| / max|Wl| per lane:", (np.abs(ql - Wl[:, order]).max(1) / np.abs(Wl).max(1)).max())
print("Q(Wl) against the LINEAR order instead:", (np.abs(ql - Wl).max(1) / np.abs(Wl).max(1)).max())
print("scale bytes W", (sc & 255)[:4], "Wl", ((sc >> 8) & 255)[:4], " exponents of max|W|", np.floor(np.log2(np.abs(W).max(1)))[:4] + 127, "max|Wl|", np.floor(np.log2(np.abs(Wl).max(1)))[:4] + 127)
np.set_printoptions(linewidth=250, precision=4, suppress=True)
err = np.abs(q - W[:, order]).max(1) / np.abs(W).max(1)
print("lanes with Q(W) error > 0.1:", np.nonzero(err > 0.1)[0])
for ln in (0, int(np.argmax(err))):
    print("lane", ln, "W (interleaved order):", W[ln, order])
    print("lane", ln, "decoded Q(W)         :", q[ln])
    print("lane", ln, "raw codes            :", unpack6(np.concatenate([wa, wb], 1))[ln])
