"""Round-3 numerics experiment (CPU, fp64 emulation): which operand formats keep sdf / normal / feature within 1e-4?

A product of a layer is emulated as  sum over the listed (weight part, activation part) pairs, accumulated in fp64
(the fp32 accumulation of the MFMA adds ~1e-7 relative and is ignored).  Schemes:
  bf3   : bf16 hi+lo both sides, 3 products (the shipped parity arithmetic)
  h3    : fp16 hi+lo both sides, 3 products
  h2a   : weights fp16 hi+lo, activations ONE fp16            (2 products)
  h2w   : weights ONE fp16, activations fp16 hi+lo            (2 products)
  h1    : one fp16 each side
  bf1   : one bf16 each side (fast mode)
Usage: python tools/experiments/r03/num_schemes.py
"""
import math, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "../../factored-neus_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "../.."))
from fneus import synth
from oracle import ref_torch as R

torch.set_default_dtype(torch.float64)

def rnd(x, dt):
    return x.to(dt).to(torch.float64)

def parts(x, dt, n):
    hi = rnd(x, dt)
    if n == 1:
        return [hi]
    return [hi, rnd(x - hi, dt)]

def prod(W, X, scheme):
    """X [M,K] @ W[O,K]^T with operand rounding."""
    dt = torch.bfloat16 if scheme.startswith("bf") else torch.float16
    if scheme in ("bf3", "h3"):
        wp, xp = parts(W, dt, 2), parts(X, dt, 2)
        return xp[0] @ wp[0].t() + xp[1] @ wp[0].t() + xp[0] @ wp[1].t()
    if scheme == "h2a":
        wp, xp = parts(W, dt, 2), parts(X, dt, 1)
        return xp[0] @ (wp[0] + wp[1]).t()
    if scheme == "h2w":
        wp, xp = parts(W, dt, 1), parts(X, dt, 2)
        return (xp[0] + xp[1]) @ wp[0].t()
    if scheme in ("h1", "bf1"):
        return rnd(X, dt) @ rnd(W, dt).t()
    if scheme == "exact":
        return X @ W.t()
    raise ValueError(scheme)

def run(x, p, layer_scheme):
    """forward + reverse sweep; layer_scheme(l, direction) -> scheme name"""
    Ws, bs = p["W"], p["b"]
    h0 = R.embed(x, 6)
    h = h0
    zs = []
    for l in range(9):
        if l == 4:
            h = torch.cat([h, h0], -1) / math.sqrt(2.0)
        z = prod(Ws[l], h, layer_scheme(l, "f")) + bs[l]
        zs.append(z)
        h = R.softplus100(z) if l < 8 else z
    out = h
    g = Ws[8][0:1, :].expand(x.shape[0], -1)
    q_skip = None
    for l in range(7, -1, -1):
        if l + 1 == 4:
            g = g / math.sqrt(2.0)
            q_skip = g[:, -39:]
            g = g[:, :-39]
        a = R.softplus100_d1(zs[l]) * g
        g = prod(Ws[l].t().contiguous(), a, layer_scheme(l, "r"))
    q = g + q_skip
    n = R.embed_jacobian_apply_T(x, q, 6)
    return out[:, :1], out[:, 1:], n

def main():
    for name, kw in [("bench net (seed 0, perturb 0.02)", dict(seed=0, perturb=0.02)),
                     ("perturb 0.1", dict(seed=3, perturb=0.1)),
                     ("warp", dict(seed=5, perturb=0.05, warp=(2, 0.15)))]:
        sd = {k: torch.from_numpy(v).double() for k, v in synth.sdf_state_dict(**kw).items()}
        p = R.sdf_params_from_state_dict(sd)
        rs = np.random.RandomState(7)
        x = torch.from_numpy(rs.uniform(-1, 1, size=(4096, 3)))
        x = x[(x.norm(dim=1) < 1.2)]
        s0, f0, n0 = run(x, p, lambda l, d: "exact")
        print(f"== {name}: {x.shape[0]} pts, |sdf| max {s0.abs().max():.3f}, |n| mean {n0.norm(dim=1).mean():.3f}, |feat| max {f0.abs().max():.3f}")
        def rep(tag, fn):
            s, f, n = run(x, p, fn)
            print(f"  {tag:34s} sdf {float((s - s0).abs().max()):.2e}  feat {float((f - f0).abs().max()):.2e}  normal {float((n - n0).abs().max()):.2e}"
                  f"   (rms sdf {float((s - s0).pow(2).mean().sqrt()):.1e} n {float((n - n0).pow(2).mean().sqrt()):.1e})")
        for sc in ("bf3", "h3", "h2a", "h2w", "h1", "bf1"):
            rep(sc, lambda l, d, sc=sc: sc)
        rep("h2w, layers 0,8 h3", lambda l, d: "h3" if l in (0, 8) else "h2w")
        rep("h2a, layers 0,8 h3", lambda l, d: "h3" if l in (0, 8) else "h2a")
        rep("h2w, layers 0,4,8 h3", lambda l, d: "h3" if l in (0, 4, 8) else "h2w")
        rep("h2w fwd, h3 reverse", lambda l, d: "h3" if d == "r" else "h2w")
        rep("h3 fwd, h2w reverse", lambda l, d: "h2w" if d == "r" else "h3")
        rep("h3 fwd, h2a reverse", lambda l, d: "h2a" if d == "r" else "h3")
        rep("h3 fwd, h1 reverse", lambda l, d: "h1" if d == "r" else "h3")

if __name__ == "__main__":
    main()
