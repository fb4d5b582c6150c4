"""Round 6 (prototype at commit beb36f6): does the fp16 matrix instruction flush subnormal operands?  The colour backward on fp16 cotangents
with the output layer's weights scaled by 2^-k: zbar_3 = W_4^T zbar_4 then sits k binades below the seed, whose largest entry the
per-sample scale puts at 2^-6; fp16's normal range ends at 2^-14, its subnormals at 2^-24."""
import os, sys
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth, pp
dev = torch.device("cuda:0"); n = 65536
g = torch.Generator(device=dev).manual_seed(9)
x = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1).contiguous()
d = torch.randn(n, 3, device=dev, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
nrm, feat, c0 = torch.randn(n, 3, device=dev, generator=g), (torch.randn(n, 256, device=dev, generator=g) * 0.3).contiguous(), torch.randn(n, 3, device=dev, generator=g)
for k in (0, 4, 8, 12, 16):
    sd = {kk: torch.from_numpy(v).clone() for kk, v in synth.color_state_dict(23).items()}
    for kk in sd:
        if kk.startswith("lin4.weight_g"): sd[kk] *= 2.0 ** -k
    net = ops.PackedNet("color", dev).load_state_dict(sd); net.pack()
    st = ops.ColStash(n, dev, 3, gprec=2)
    rgb = ops.color_fwd(net.blob, n, 3, nrm, feat, st, True, pts=x, dirs=d)
    w16 = ops.h16_blob(net, 1)
    os.environ["FNEUS_COLB_XHI"] = "0"; ref, _ = ops.color_bwd(net.blob, n, 3, c0, rgb, st); ref = ref.clone()
    os.environ["FNEUS_COLB_XHI"] = "1"; out, _ = ops.color_bwd(net.blob, n, 3, c0, rgb, st, w16=w16); torch.cuda.synchronize()
    print("output-layer weights x 2^-%d: d_feat of the fp16 chain against the hi + lo chain: L2 %.2e" % (k, (out - ref).norm().item() / ref.norm().item()))
