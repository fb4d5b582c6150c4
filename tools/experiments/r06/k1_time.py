"""time of the shipped two-pass K1 (parity mode, prec 3; and plain bf16, prec 1) per launch at 65 536 and 1 M points -- for builds
with parts of the pass compiled out (tools/experiments/r06/k1_overlap.sh; results of such builds are wrong, times only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, synth
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()})
net.pack()
out = []
for n in (65536, 1 << 20):
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.rand(n, 3, device=dev, generator=g) * 2.2 - 1.1).contiguous()
    for prec in (3, 1):
        fn = lambda: ops.sdf_fwd(net.blob, n, prec, pts=x)
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(f"n {n} prec {prec}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
print("  ".join(out), flush=True)
