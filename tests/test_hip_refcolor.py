"""GPU parity of the surface head RefColor (reference models/fields.py:271-335) on the fused colour-network kernels:
outputs and gradients (inputs and parameters) against fp64 autograd of the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _inputs(m, seed):
    rs = np.random.RandomState(seed)
    pts = T(rs.uniform(-1.0, 1.0, size=(m, 3)).astype(np.float32))
    d = T(rs.standard_normal((m, 3)).astype(np.float32))
    d = d / d.norm(dim=-1, keepdim=True)
    n = T((rs.standard_normal((m, 3)) * 1.3).astype(np.float32))        # raw SDF gradients are not unit length
    feat = T((rs.standard_normal((m, 256)) * 0.3).astype(np.float32))
    return pts, feat, d, n


def _module(seed, prec):
    from fneus import synth
    from models.fields import RefColor
    sd = {k: T(v) for k, v in synth.refcolor_state_dict(seed).items()}
    mod = RefColor()
    mod.load_state_dict(sd)
    mod.to(torch.device("cuda:0"))
    mod.set_precision(prec)
    mod.refresh()
    return mod, sd


@pytest.mark.parametrize("prec,tol", [(3, 1e-4), (1, 3e-2)])
def test_refcolor_forward_matches_oracle(prec, tol):
    from oracle import ref_torch as R
    dev = torch.device("cuda:0")
    m = 777                                          # ragged: not a multiple of the 32-sample wave tile
    pts, feat, d, n = _inputs(m, 3)
    mod, sd = _module(31, prec)
    ref = R.refcolor_forward(pts.double(), feat.double(), d.double(), n.double(), {k: v.double() for k, v in sd.items()})
    with torch.no_grad():
        out = mod(pts.to(dev), feat.to(dev), d.to(dev), n.to(dev))
    for k in ("rgb", "specular_rgb", "diffuse_rgb"):
        err = (out[k].cpu().double() - ref[k]).abs().max().item()
        print(f"  refcolor prec={prec} {k}: max abs err {err:.3e}")
        assert err <= tol, k


# gprec 3: hi + lo planes (fp32-accurate weight gradients); gprec 1 (training default): bf16 planes, every product of the
# weight-gradient GEMM carries 2^-9 rounding (random cotangents: no cancellation); input gradients do not depend on it
@pytest.mark.parametrize("prec,gprec,tol,gtol,wtol", [(3, 3, 1e-4, 5e-4, 5e-4), (3, 1, 1e-4, 5e-4, 6e-3), (1, 1, 3e-2, 1e-1, 1e-1)])
def test_refcolor_gradients_match_oracle(prec, gprec, tol, gtol, wtol):
    from oracle import ref_torch as R
    dev = torch.device("cuda:0")
    m = 1024
    pts, feat, d, n = _inputs(m, 4)
    rs = np.random.RandomState(9)
    mod, sd = _module(32, prec)
    mod.set_gradient_precision(gprec)
    # ---- oracle, fp64 autograd; the loss goes through the heads before the (piecewise) sRGB transfer and clipping
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    n64, f64 = n.double().requires_grad_(True), feat.double().requires_grad_(True)
    ref = R.refcolor_forward(pts.double(), f64, d.double(), n64, sd64)
    c = {k: T(rs.standard_normal((m, 3)).astype(np.float32)) for k in ("rgb", "specular_rgb", "diffuse_rgb")}
    # samples whose ReLU pre-activations sit within rounding of zero have no well-defined derivative: zero cotangent
    with torch.no_grad():
        def zmin(x, keys):
            z = []
            for wk, bk in keys:
                x = x @ sd64[wk].t() + sd64[bk]
                z.append(x.abs().min(dim=1)[0])
                x = torch.relu(x)
            return torch.stack(z).min(dim=0)[0]
        n_enc, ref_enc = R.embed(n.double(), 4), R.embed(R.reflect(-d.double(), R.l2_normalize(n.double())), 4)
        z1 = zmin(torch.cat([pts.double(), n_enc, feat.double()], -1), [(f"net_cd.{i}.weight", f"net_cd.{i}.bias") for i in (0, 2, 4, 6)])
        z2 = zmin(torch.cat([n.double(), pts.double(), ref_enc, feat.double()], -1),
                  [(f"viewdir_mlp.{i}.weight", f"viewdir_mlp.{i}.bias") for i in range(4)])
        ok = (torch.minimum(z1, z2) > (3e-6 if prec == 3 else 0.0)).float()[:, None]
    loss_ref = sum((ref[k] * (c[k] * ok).double()).sum() for k in c)
    loss_ref.backward()
    # ---- HIP
    nd, fd = n.to(dev).requires_grad_(True), feat.to(dev).requires_grad_(True)
    out = mod(pts.to(dev), fd, d.to(dev), nd)
    loss = sum((out[k] * (c[k] * ok).to(dev)).sum() for k in c)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= tol * max(1.0, abs(loss_ref.item())) * 30
    e_n, e_f = rel_err(nd.grad, n64.grad), rel_err(fd.grad, f64.grad)
    print(f"  refcolor prec={prec} d_normal rel {e_n:.3e} d_feat rel {e_f:.3e}")
    assert e_n <= gtol and e_f <= gtol
    worst = 0.0
    for name, p in mod.named_parameters():
        e = rel_err(p.grad, sd64[name].grad)
        print(f"  refcolor prec={prec} gprec={gprec} d {name}: rel {e:.3e}")
        worst = max(worst, e)
    assert worst <= wtol


def test_refcolor_state_dict_roundtrip_through_flat_buffers():
    """parameters alias the flat device buffers after refresh(); state_dict / load_state_dict must still see them"""
    mod, sd = _module(33, 3)
    got = mod.state_dict()
    assert list(got.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(got[k].cpu(), sd[k]), k
    with torch.no_grad():
        mod.net_cd[0].weight.mul_(0.5)
    assert torch.allclose(mod._cd.net.raw_views(mod._cd.net.raw)[0]["weight"].cpu(), sd["net_cd.0.weight"] * 0.5)
