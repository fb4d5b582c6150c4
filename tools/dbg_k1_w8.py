#!/usr/bin/env python3
"""K1 on 8-wave workgroups (sdf_w8_kernels.hip) against the round-2 kernels: values (vs K2's sdf on the same points and vs
the other K1 kernels) and time.  Usage: python tools/dbg_k1_w8.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import torch
from fneus import ops, synth
dev = torch.device("cuda:0")
net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(20).items()}); net.pack()

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def setenv(**kw):
    for k, v in kw.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)

for prec in (3, 1):
    for n in ([int(a) for a in sys.argv[1:]] or (2048, 8192, 16384, 32768, 65536, 40003, 1048576)):
        x = (torch.rand(n, 3, device=dev) * 2 - 1).contiguous()
        st = ops.SdfStash(n, dev, prec, False)
        ref, _, _ = ops.sdf_fwd_grad(net.blob, n, prec, st, False, pts=x)
        big = n >= 1024 * 32
        variants = [("r02", dict(FNEUS_K1_W8_BIG=0, FNEUS_K1_W8_SMALL=0))]
        if big:
            variants += [("w8 hb4", dict(FNEUS_K1_W8_BIG=4)), ("s8 hb2", dict(FNEUS_K1_W8_BIG=22)), ("p2", dict(FNEUS_K1_W8_BIG=3)), ("p2 8w", dict(FNEUS_K1_W8_BIG=31)), ("p2h", dict(FNEUS_K1_W8_BIG=32))]
        else:
            variants += [("w8 hb1", dict(FNEUS_K1_W8_SMALL=1)), ("w8p", dict(FNEUS_K1_W8_SMALL=2))]
        line = f"prec {prec} n={n:8d}:"
        for name, env in variants:
            setenv(**env)
            out = ops.sdf_fwd(net.blob, n, prec, pts=x)
            same = all(torch.equal(ops.sdf_fwd(net.blob, n, prec, pts=x), out) for _ in range(5))
            t = timeit(lambda: ops.sdf_fwd(net.blob, n, prec, pts=x))
            line += f"  | {name}: {t:7.1f} us, |d K2| {(out - ref).abs().max().item():.1e}{'' if same else ' NOT REPEATABLE'}"
        print(line, flush=True)
setenv(FNEUS_K1_W8_BIG=None, FNEUS_K1_W8_SMALL=None)
