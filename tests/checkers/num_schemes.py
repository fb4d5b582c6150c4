"""Round-3 numerics experiment (CPU, fp64 emulation): which operand formats keep sdf / normal / feature within 1e-4?

A product of a layer is emulated as  sum over the listed (weight part, activation part) pairs, accumulated in fp64
(the fp32 accumulation of the MFMA adds ~1e-7 relative and is ignored).  Schemes:
  bf3   : bf16 hi+lo both sides, 3 products (the shipped parity arithmetic)
  h3    : fp16 hi+lo both sides, 3 products
  h2a   : weights fp16 hi+lo, activations ONE fp16            (2 products)
  h2w   : weights ONE fp16, activations fp16 hi+lo            (2 products)
  h1    : one fp16 each side
  bf1   : one bf16 each side (fast mode)
  h6 / h8 / h4 / bf6 (round 5): ONE 16-bit product hi.hi (fp16; bf6: bf16) + the two cross terms hi.lo, lo.hi from block-scaled
          low-precision operands (MX: one power-of-two scale per 32 consecutive k; fp6 e2m3, fp8 e4m3, fp4 e2m1) -- the form
          v_mfma_scale_f32_32x32x64_f8f6f4 multiplies at 4x (fp6 / fp4) or 2x (fp8) the bf16 rate: 1.5 / 2 MFMA-times per product
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

NO_SAT = True


def mx_quant(x, fmt):
    """block-scaled quantisation along the last axis (blocks of 32, zero padded): shared scale 2^(floor(log2 max) - emax), elements
    rounded to nearest on the format's grid, saturating (OCP MX)"""
    mbits, emax, top = {"fp6": (3, 2, 7.5), "fp8": (3, 8, 448.0), "fp4": (1, 2, 6.0)}[fmt]
    K = x.shape[-1]
    pad = (-K) % 32
    xp = torch.nn.functional.pad(x, (0, pad)).reshape(x.shape[:-1] + (-1, 32))
    mx = xp.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.floor(torch.log2(mx)) - emax)
    if NO_SAT:          # our own conversion picks the scale: the smallest power of two that does not saturate the block's maximum
        half_top_step = {"fp6": 0.25, "fp8": 16.0, "fp4": 1.0}[fmt]
        scale = torch.where(mx / scale >= top + half_top_step, scale * 2.0, scale)
    v = xp / scale
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-300))).clamp_min(0.0 if fmt != "fp8" else -6.0)   # subnormals below 2^emin
    step = torch.exp2(e - mbits)
    q = (torch.round(v / step) * step).clamp(-top, top)
    return (q * scale).reshape(x.shape[:-1] + (-1,))[..., :K]


def prod(W, X, scheme):
    """X [M,K] @ W[O,K]^T with operand rounding."""
    if scheme in ("h6", "h8", "h4", "bf6"):
        dt = torch.bfloat16 if scheme == "bf6" else torch.float16
        fmt = {"h6": "fp6", "bf6": "fp6", "h8": "fp8", "h4": "fp4"}[scheme]
        wh, xh = rnd(W, dt), rnd(X, dt)
        return xh @ wh.t() + mx_quant(X - xh, fmt) @ mx_quant(W, fmt).t() + mx_quant(X, fmt) @ mx_quant(W - wh, fmt).t()
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
        for sc in ("bf3", "h3", "h6", "h8", "bf6", "h4", "h2a", "h2w", "h1", "bf1"):
            rep(sc, lambda l, d, sc=sc: sc)
        rep("h6 fwd, bf3 reverse", lambda l, d: "bf3" if d == "r" else "h6")
        rep("bf3 fwd, h6 reverse", lambda l, d: "h6" if d == "r" else "bf3")
        rep("h2w, layers 0,8 h3", lambda l, d: "h3" if l in (0, 8) else "h2w")
        rep("h2a, layers 0,8 h3", lambda l, d: "h3" if l in (0, 8) else "h2a")
        rep("h2w, layers 0,4,8 h3", lambda l, d: "h3" if l in (0, 4, 8) else "h2w")
        rep("h2w fwd, h3 reverse", lambda l, d: "h3" if d == "r" else "h2w")
        rep("h3 fwd, h2w reverse", lambda l, d: "h2w" if d == "r" else "h3")
        rep("h3 fwd, h2a reverse", lambda l, d: "h2a" if d == "r" else "h3")
        rep("h3 fwd, h1 reverse", lambda l, d: "h1" if d == "r" else "h3")

if __name__ == "__main__":
    main()
