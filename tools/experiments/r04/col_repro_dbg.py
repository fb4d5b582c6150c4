import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np, torch
from fneus import ops, synth
dev = torch.device("cuda:0")
n = 65536
for prec in (1, 3):
    cnet = ops.PackedNet("color", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.color_state_dict(21).items()})
    cnet.pack()
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1).contiguous()
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1).contiguous()
    nrm = torch.randn(n, 3, device=dev, generator=g)
    feat = (torch.randn(n, 256, device=dev, generator=g) * 0.3).contiguous()
    c = torch.randn(n, 3, device=dev, generator=g)
    st = ops.ColStash(n, dev, prec)
    def run():
        rgb = ops.color_fwd(cnet.blob, n, prec, nrm, feat, st, True, pts=x, dirs=d)
        d_feat, d_normal = ops.color_bwd(cnet.blob, n, prec, c, rgb, st)
        torch.cuda.synchronize()
        return [t.clone() for t in (rgb, d_feat, d_normal, st.u.view(torch.int16), st.zbar.view(torch.int16), st.mask, st.side.view(torch.int16))]
    ref = run()
    bad = {}
    for it in range(60):
        for a, b, name in zip(run(), ref, ("rgb", "d_feat", "d_normal", "u", "zbar", "mask", "side")):
            if not torch.equal(a, b):
                diff = (a != b)
                idx = diff.nonzero()
                bad.setdefault(name, []).append((it, int(diff.sum()), idx[0].tolist(), idx[-1].tolist()))
    print("prec", prec, "COL_P2", os.environ.get("FNEUS_COL_P2", "1"), {k: (len(v), v[:3]) for k, v in bad.items()})
