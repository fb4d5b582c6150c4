"""Properties that do not depend on the batch size, checked at the size BASELINE.json quotes (512 rays x 128 samples),
where the oracle is too slow to be the checker: ray-permutation equivariance, compositing invariants, linearity of the
backward in its cotangent, repeatability.  Plus the ragged / degenerate launch sizes of the per-sample kernels against
the oracle (1, 31, 33 samples; empty launches)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def _trainer(seed=11):
    from fneus import ops
    from fneus.trainer import Stage1Trainer, WMASK_MODEL
    conf = copy.deepcopy(WMASK_MODEL)
    conf["neus_renderer"]["perturb"] = 0.0
    return Stage1Trainer(DEV, model_conf=conf, prec=ops.PREC_PARITY, seed=seed, use_graph=False)


def test_full_size_render_is_equivariant_under_ray_permutation_and_keeps_the_compositing_invariants():
    from fneus.trainer import synthetic_batches
    tr = _trainer()
    data = synthetic_batches(1, 512, DEV, seed0=31)[0]
    perm = torch.randperm(512, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    a = tr.render_only(data)
    b = tr.render_only(data[perm].contiguous())
    for k in ("color_fine", "weights", "weight_sum", "gradients", "cdf_fine", "inside_sphere", "surface_color", "sdf_mask",
              "_z_vals", "_sdf"):
        va = a[k].reshape(512, -1)[perm]
        vb = b[k].reshape(512, -1)
        assert torch.equal(va, vb), k              # every kernel on the path works ray by ray / tile by tile: bit-exact
    w, z = a["weights"], a["_z_vals"]
    assert w.shape == (512, 128) and (w >= 0).all() and (a["weight_sum"] <= 1.0 + 1e-5).all()
    assert (z[:, 1:] >= z[:, :-1]).all()                                      # merged depths stay sorted
    ins = a["inside_sphere"]
    assert ((ins == 0) | (ins == 1)).all()
    assert ((a["cdf_fine"] >= 0) & (a["cdf_fine"] <= 1)).all()
    sc, m = a["surface_color"], a["sdf_mask"]
    assert (sc[~m] == 1.0).all()                                              # rows outside sdf_mask stay 1.0 (renderer.py:280-282)
    assert int(m.sum()) > 50                                                  # ... and the branch is exercised
    again = tr.render_only(data)
    for k in ("color_fine", "weights", "gradients", "surface_color"):
        assert torch.equal(a[k], again[k]), k                                 # repeatable bit for bit


def test_full_size_backward_is_linear_in_its_cotangent():
    from fneus import ops, synth
    n = 65536
    net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(5).items()})
    net.pack()
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.rand(n, 3, device=DEV, generator=g) * 2 - 1).contiguous()
    stash = ops.SdfStash(n, DEV, 3, train=True)
    ops.sdf_fwd_grad(net.blob, n, 3, stash, True, pts=x)
    bufs = ops.SdfBwdBufs(n, DEV, 3)
    c_s = torch.randn(n, device=DEV, generator=g)
    c_f = torch.randn(n, 256, device=DEV, generator=g) * 0.05
    c_n = torch.randn(n, 3, device=DEV, generator=g)

    def grads(scale):
        ops.sdf_bwd(net.blob, n, 3, stash, bufs, (c_s * scale).contiguous(), (c_f * scale).contiguous(), (c_n * scale).contiguous(), pts=x)
        grad = torch.zeros(net.n_params, dtype=torch.float32, device=DEV)
        ops.sdf_dw_jobs(net, stash, bufs, grad, n).run(n, 3)
        torch.cuda.synchronize()
        return grad

    g1, g4 = grads(1.0), grads(4.0)
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    rel = ((g4 - 4.0 * g1).norm() / g4.norm()).item()
    print(f"  dW(4 c) vs 4 dW(c) at N = 65536: relative difference {rel:.2e}")
    assert rel <= 2e-6            # powers of two scale exactly; what is left is the order of the fp32 atomics


@pytest.mark.parametrize("n", [1, 31, 33])
def test_ragged_launch_sizes_match_the_oracle(n):
    from fneus import ops, synth
    from oracle import ref_torch as R
    sd = {k: T(v) for k, v in synth.sdf_state_dict(20).items()}
    p = R.sdf_params_from_state_dict(sd)
    p64 = {"W": [w.double() for w in p["W"]], "b": [b.double() for b in p["b"]], "scale": 1.0}
    net = ops.PackedNet("sdf", DEV).load_state_dict(sd)
    net.pack()
    rs = np.random.RandomState(n)
    x = T(rs.uniform(-1, 1, size=(n, 3)).astype(np.float32))
    sdf_r, feat_r, nrm_r, _ = R.sdf_value_feature_normal(x.double(), p64)
    xd = x.to(DEV).contiguous()
    guard = torch.full((n + 64,), 7.0, device=DEV)                             # nothing may be written behind row n
    ops.sdf_fwd(net.blob, n, 3, pts=xd, out=guard[:n])
    assert (guard[n:] == 7.0).all()
    assert (guard[:n].cpu().double() - sdf_r[:, 0]).abs().max().item() <= 1e-4
    stash = ops.SdfStash(n, DEV, 3, train=True)
    sdf, feat, nrm = ops.sdf_fwd_grad(net.blob, n, 3, stash, True, pts=xd)
    assert (sdf.cpu().double() - sdf_r[:, 0]).abs().max().item() <= 1e-4
    assert (feat.cpu().double() - feat_r).abs().max().item() <= 1e-4
    assert (nrm.cpu().double() - nrm_r).abs().max().item() <= 1e-4
    # colour network on the same ragged tile
    csd = {k: T(v) for k, v in synth.color_state_dict(21).items()}
    cnet = ops.PackedNet("color", DEV).load_state_dict(csd)
    cnet.pack()
    d = T(rs.standard_normal((n, 3)).astype(np.float32))
    d = d / d.norm(dim=-1, keepdim=True)
    rgb_ref = R.color_forward(x.double(), nrm_r, d.double(), feat_r, {k: [t.double() for t in v] for k, v in
                                                                        R.color_params_from_state_dict(csd).items()})
    rgb = ops.color_fwd(cnet.blob, n, 3, nrm, feat, None, False, pts=xd, dirs=d.to(DEV).contiguous())
    assert (rgb.cpu().double() - rgb_ref).abs().max().item() <= 1e-4


def test_empty_launches_are_no_ops():
    from fneus import ops, synth
    net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(20).items()})
    net.pack()
    x = torch.zeros(0, 3, device=DEV)
    assert ops.sdf_fwd(net.blob, 0, 3, pts=x).shape == (0,)
    stash = ops.SdfStash(32, DEV, 3, train=True)
    sdf, feat, nrm = ops.sdf_fwd_grad(net.blob, 0, 3, stash, True, pts=x)
    assert sdf.shape == (0,) and feat.shape == (0, 256) and nrm.shape == (0, 3)
    torch.cuda.synchronize()


def test_stage2_entry_util_matches_the_oracle_composition():
    """NeuSRenderer.lvis_mateIllu_render_util (renderer.py:503-564, the entry of the stage-2/3 renderers): K1 + K6 only"""
    from fneus import synth
    from fneus.trainer import synthetic_batches
    from oracle import ref_torch as R
    tr = _trainer(seed=11)
    data = synthetic_batches(1, 96, DEV, seed0=5)[0]
    o, d = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    near, far = R.near_far_from_sphere(o.cpu(), d.cpu())
    out = tr.renderer.lvis_mateIllu_render_util(o, d, near.to(DEV), far.to(DEV))
    assert out["n_samples"] == 128 and out["mid_z_vals"].shape == (96, 128) and out["sdf"].shape == (96 * 128, 1)
    sdf_p = R.sdf_params_from_state_dict({k: T(v) for k, v in synth.sdf_state_dict(11).items()})
    sdf_fn = lambda p: R.sdf_only(p, sdf_p)
    z = R.hierarchical_z(o.cpu(), d.cpu(), R.initial_z_vals(near, far, 64), sdf_fn, 64, 4)
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 2.0 / 64)], -1)
    mid_ref = z + 0.5 * dists
    mid = out["mid_z_vals"].cpu()
    # the inverse-CDF sampler is ill-conditioned where the pdf is flat: compare depths loosely ...
    frac_off = ((mid - mid_ref).abs() > 2e-3).float().mean().item()
    assert frac_off < 0.02, frac_off
    # ... and everything computed FROM the depths exactly, at the kernel's own depths
    pts = o.cpu()[:, None, :] + d.cpu()[:, None, :] * mid[..., None]
    sdf_ref = R.sdf_only(pts.reshape(-1, 3), sdf_p)
    assert (out["sdf"].cpu() - sdf_ref).abs().max().item() <= 1e-4
    inside_ref = (torch.linalg.norm(pts, dim=-1) < 1.0).any(dim=-1)
    assert torch.equal(out["inside_sphere_mask"].cpu(), inside_ref)


def test_overwritten_stash_is_detected():
    """two differentiable SDF calls of the same size before the first backward share one stash: the first backward must
    refuse to run on the second call's activations instead of silently returning wrong gradients"""
    from models.fields import SDFNetwork
    net = SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                     geometric_init=True, weight_norm=True).to(DEV)
    a = torch.rand(64, 3, device=DEV) - 0.5
    b = torch.rand(64, 3, device=DEV) - 0.5
    ga = net.gradient(a)
    gb = net.gradient(b)
    gb.sum().backward()                      # the latest call: fine
    with pytest.raises(RuntimeError, match="overwritten"):
        ga.sum().backward()


def test_frozen_network_packs_again_only_after_a_parameter_changed(monkeypatch):
    """stages 2 / 3 refresh their frozen networks every step: the pack launches are skipped while no parameter has been written
    (address + version of every parameter), and come back with the first in-place write; a trained network always packs"""
    from fneus import ops
    from fneus.autograd import RaySamples
    from models.fields import SDFNetwork
    net = SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                     geometric_init=True, weight_norm=True).to(DEV)
    calls = []
    real = ops.PackedNet.pack
    monkeypatch.setattr(ops.PackedNet, "pack", lambda self: (calls.append(1), real(self))[1])
    pts = RaySamples(pts=(torch.rand(256, 3, device=DEV) - 0.5).contiguous())
    net.refresh()
    net.refresh()
    assert len(calls) == 2                                  # trained parameters: every refresh packs
    for p in net.parameters():
        p.requires_grad_(False)
    net.refresh()
    y0 = net.sdf_samples(pts).clone()
    net.refresh()
    net.refresh()
    assert len(calls) == 3                                  # frozen and unchanged: packed once
    with torch.no_grad():
        net.lin8.bias.add_(0.25)
    net.refresh()
    assert len(calls) == 4
    y1 = net.sdf_samples(pts)
    assert (y1 - y0 - 0.25).abs().max().item() <= 1e-5      # lin8's first output row is the sdf
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net.refresh()
    assert len(calls) == 5                                  # a state_dict load writes every parameter


def _grads_of(mod):
    return torch.cat([p.grad.detach().reshape(-1).clone() for p in mod.parameters()])


def test_64_sample_workgroups_equal_32_sample_workgroups(monkeypatch):
    """chip-filling launches (>= 1024 tiles) run kernels whose workgroups carry two 32-sample tiles through one pass over the
    weight fragments (colour network, background NeRF; FNEUS_COL_HB / FNEUS_K7_HB = 1 select the 32-sample kernels, read at
    every call).  Same arithmetic per sample: outputs and input gradients agree to rounding-order noise, weight gradients to
    the summation order of the GEMM's atomics.  Ragged size with an odd tile count."""
    from fneus import ops, synth
    from fneus.autograd import RaySamples
    from models.fields import SDFNetwork, RenderingNetwork, NeRF
    from fneus.trainer import WMASK_MODEL
    dev = torch.device("cuda:0")
    n = 40003
    g = torch.Generator().manual_seed(3)
    T = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
    # ---- colour network (K4) ----
    sdf = SDFNetwork(**WMASK_MODEL["sdf_network"])
    col = RenderingNetwork(**WMASK_MODEL["rendering_network"])
    sdf.load_state_dict(T(synth.sdf_state_dict(1)))
    col.load_state_dict(T(synth.color_state_dict(2)))
    sdf.to(dev), col.to(dev)
    pts = (torch.rand(n, 3, generator=g) * 1.6 - 0.8).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    cot = torch.randn(n, 3, generator=g).to(dev)

    def run_color(hb, p2=0):
        monkeypatch.setenv("FNEUS_COL_HB", str(hb))
        monkeypatch.setenv("FNEUS_COL_P2", str(p2))
        for p in list(sdf.parameters()) + list(col.parameters()):
            p.grad = None
        s = RaySamples(pts=pts, dirs=dirs)
        sdf.refresh(), col.refresh()
        _, feat, normal = sdf.value_feature_normal(s, True)
        rgb = col.color_samples(s, normal, feat, sdf, True)
        (rgb * cot).sum().backward()
        return rgb.detach().clone(), _grads_of(col), _grads_of(sdf)

    rgb2, gc2, gs2 = run_color(2)
    rgb1, gc1, gs1 = run_color(1)
    assert (rgb2 - rgb1).abs().max().item() <= 1e-6
    for a, b, name in ((gc2, gc1, "colour"), (gs2, gs1, "sdf")):
        rel = ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
        assert rel <= 2e-4, (name, rel)
    # the forward in the two-pass pipelined form (csrc/color_p2_kernels.hip, the default for such launches; FNEUS_COL_P2=0 above):
    # same hidden layers, the 256 -> 3 output layer as vector dot products; its planes and ReLU masks feed the same backward
    rgbp, gcp, gsp = run_color(2, p2=1)
    assert (rgbp - rgb2).abs().max().item() <= 2e-6
    for a, b, name in ((gcp, gc2, "colour"), (gsp, gs2, "sdf")):
        rel = ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
        assert rel <= 2e-4, (name, rel)
    # ---- background NeRF (K7) ----
    nerf = NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True)
    nerf.load_state_dict(T(synth.nerf_state_dict(3)))
    nerf.to(dev)
    p4 = torch.cat([torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1), torch.rand(n, 1, generator=g) * 0.98 + 0.02], -1).to(dev)
    ca, cr = torch.randn(n, 1, generator=g).to(dev), torch.randn(n, 3, generator=g).to(dev)

    def run_nerf(hb):
        monkeypatch.setenv("FNEUS_K7_HB", str(hb))
        monkeypatch.setenv("FNEUS_NERF_XHI", "0")           # (the same arithmetic on both sides: hi + lo cotangents inside the chain;
                                                            #  the bf16-cotangent form of the 64-sample kernel: tests/test_hip_nerf.py)
        for p in nerf.parameters():
            p.grad = None
        nerf.refresh()
        a, rgb = nerf(p4, dirs)
        ((a * ca).sum() + (rgb * cr).sum()).backward()
        return a.detach().clone(), rgb.detach().clone(), _grads_of(nerf)

    a2, r2, g2 = run_nerf(2)
    a1, r1, g1 = run_nerf(1)
    scale = max(a1.abs().max().item(), 1.0)
    assert (a2 - a1).abs().max().item() <= 2e-6 * scale and (r2 - r1).abs().max().item() <= 2e-6 * max(r1.abs().max().item(), 1.0)
    rel = ((g2 - g1).abs().max() / (g1.abs().max() + 1e-12)).item()
    assert rel <= 2e-4, rel


@pytest.mark.parametrize("train,gprec,prec", [(True, 1, 3), (True, 3, 3), (False, 1, 3), (True, 1, 1)])
def test_k2_as_two_launches_equals_the_fused_kernel(monkeypatch, train, gprec, prec):
    """chip-filling launches of K2 (>= 1024 tiles) run the forward chain in the two-pass pipelined form with the stash written
    on the way, then the reverse sweep as a launch of its own (csrc/sdf_p2_train_kernels.hip; FNEUS_K2_P2=0 keeps the fused
    32-sample kernel, read at every call).  Same operands and summation order per accumulator: the feature rows and the
    forward planes are bit-identical, sdf differs by the order of one 256-term sum (the sdf row is a vector dot product
    there), sigma' by one step of its 16-bit code, and what the reverse sweep derives from sigma' by that step.  Ragged size
    with an odd tile count; the tiles allocated beyond the samples stay zero."""
    from fneus import ops, synth
    n = 40003
    net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(20).items()})
    net.pack()
    x = (torch.rand(n, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)) * 2 - 1).contiguous()

    def run(p2):
        monkeypatch.setenv("FNEUS_K2_P2", str(p2))
        st = ops.SdfStash(n, DEV, prec, train, gprec)
        for t in (st.h, st.a, st.feat):
            if t is not None:
                t.zero_()
        st.ps.zero_()
        out = ops.sdf_fwd_grad(net.blob, n, prec, st, train, pts=x)
        torch.cuda.synchronize()
        return out, st

    (sdf0, feat0, nrm0), s0 = run(0)
    (sdf1, feat1, nrm1), s1 = run(1)
    fast = prec == 1        # bf16 mode (not a parity mode): the sdf row is an fp32 dot product here, a bf16 MFMA there
    assert (sdf1 - sdf0).abs().max().item() <= (5e-3 if fast else 1e-5)
    assert torch.equal(feat1, feat0)
    assert (nrm1 - nrm0).abs().max().item() <= (1e-2 if fast else 5e-5)
    for l in range(8):
        assert (s1.sigma(l) - s0.sigma(l)).abs().max().item() <= 2.1 / 65535.0, l
    if train:
        assert torch.equal(s1.pe, s0.pe) and torch.equal(s1.feat, s0.feat)
        for l in range(8):
            F = 14 if l == 3 else 16                        # layer 3 has 7 output tiles
            assert torch.equal(s1.h[:, l, :, :F], s0.h[:, l, :, :F]), l
            a1, a0 = s1.plane(s1.a[:, :, :, :F], l), s0.plane(s0.a[:, :, :, :F], l)
            assert (a1 - a0).abs().max().item() <= (8e-3 if gprec == 1 else 1e-4) * max(a0.abs().max().item(), 1e-6), l
        if s1.h.shape[2] > s1.tiles:                        # an allocated tile without samples
            assert float(s1.h[:, :, s1.tiles:].float().abs().max()) == 0.0


def test_batch_without_any_surface_hit():
    """when no ray of a batch hits a surface the reference skips the RefColor branch (renderer.py:296:
    its parameters get grad None and Adam skips them); here the branch runs at fixed shape with zero weights, so the RefColor
    parameters receive exactly ZERO gradients: on a fresh optimiser they do not move at all (zero moments -> zero update),
    afterwards their moments decay.  The step stays finite and the other networks train."""
    from fneus import synth
    from fneus.trainer import Stage1Trainer
    dev = torch.device("cuda:0")
    # (seed 0: with some initialisations EVERY surface colour starts clipped at 1 -- spec + diffuse ~ 0.5 + 0.5 -- and the
    # RefColor gradient is zero on hitting batches too, for the reference as for us)
    tr = Stage1Trainer(dev, seed=0, use_graph=False)
    miss = torch.from_numpy(synth.ray_batch(512, seed=5, n_miss=512)).to(dev)          # every ray passes outside the unit sphere
    before = {k: v.detach().clone() for k, v in tr.refColor_network.state_dict().items()}
    sdf_before = tr.sdf_network.lin0.weight_v.detach().clone()
    losses = tr.train_step(miss)
    assert all(bool(torch.isfinite(v)) for v in losses.values() if torch.is_tensor(v))
    out = tr.render_only(miss)
    assert not bool(out["sdf_mask"].any())
    for k, v in tr.refColor_network.state_dict().items():
        assert torch.equal(v, before[k]), k                                             # zero gradient, zero moments: no update
    assert not torch.equal(tr.sdf_network.lin0.weight_v.detach(), sdf_before)           # eikonal / mask terms still train the SDF
    # a hitting batch next: everything trains
    hit = torch.from_numpy(synth.ray_batch(512, seed=6)).to(dev)
    tr.train_step(hit)
    moved = max((v - before[k]).abs().max().item() for k, v in tr.refColor_network.state_dict().items())
    assert moved > 0.0


@pytest.mark.parametrize("nh", [4, 2])
@pytest.mark.parametrize("train,gprec,prec,n", [(True, 1, 3, 40003), (True, 3, 3, 40003), (False, 1, 3, 40003), (True, 1, 1, 40003),
                                                (True, 1, 3, 65536), (True, 1, 3, 40067)])
def test_k2_reverse_sweep_r8_equals_the_4_wave_kernel(monkeypatch, train, gprec, prec, n, nh):
    """the reverse sweep of K2 on resident-weight 8-wave workgroups (csrc/sdf_r8_kernels.hip, r8_engine.h; FNEUS_K2_REV8=0 keeps
    the 4-wave kernel of sdf_kernels.hip, read at every call): same sigma' blocks, same operands and the same summation order
    per accumulator.  What differs is where hipcc contracts the lo part of a hi / lo split (`x - hi`) with the product that
    formed x -- the rounding of a lo fragment's last bit (2^-17 of the value): the normals agree to 2e-5 (parity mode), the a_l
    planes to a step of their format, and the kernel is bit-reproducible.  64- and 128-sample workgroups (FNEUS_R8_NH); ragged
    sizes with an odd tile count (40 003) and with a last 128-sample group of which only two tiles are allocated (40 067)."""
    from fneus import ops, synth
    monkeypatch.setenv("FNEUS_R8_NH", str(nh))
    net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(21).items()})
    net.pack()
    x = (torch.rand(n, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) * 2 - 1).contiguous()

    def run(r8):
        monkeypatch.setenv("FNEUS_K2_REV8", str(r8))
        st = ops.SdfStash(n, DEV, prec, train, gprec)
        if st.a is not None:
            st.a.zero_()
        out = ops.sdf_fwd_grad(net.blob, n, prec, st, train, pts=x)
        torch.cuda.synchronize()
        return out, st

    (sdf0, feat0, nrm0), s0 = run(0)
    (sdf1, feat1, nrm1), s1 = run(1)
    (sdf2, feat2, nrm2), s2 = run(1)
    fast = prec == 1
    assert torch.equal(sdf1, sdf0) and torch.equal(feat1, feat0)
    assert torch.isfinite(nrm1).all()
    assert (nrm1 - nrm0).abs().max().item() <= (2e-2 if fast else 2e-5)
    assert torch.equal(nrm2, nrm1)
    if train:
        assert torch.equal(s2.a, s1.a)
        for l in range(8):
            F = 14 if l == 3 else 16                        # layer 3 has 7 output tiles
            a1, a0 = s1.plane(s1.a[:, :, :, :F], l), s0.plane(s0.a[:, :, :, :F], l)
            assert (a1 - a0).abs().max().item() <= (8e-3 if gprec == 1 else 1e-4) * max(a0.abs().max().item(), 1e-6), l
        if s1.a.shape[2] > s1.tiles:                        # an allocated tile without samples stays zero
            assert float(s1.a[:, :, s1.tiles:].float().abs().max()) == 0.0


@pytest.mark.parametrize("nh", [4, 2])
@pytest.mark.parametrize("gprec,prec,n", [(1, 3, 40003), (3, 3, 40003), (1, 1, 40003), (1, 3, 65536), (1, 3, 40067)])
def test_k3_r8_equals_the_4_wave_kernel(monkeypatch, gprec, prec, n, nh):
    """K3 on resident-weight 8-wave workgroups (csrc/sdf_r8_kernels.hip; FNEUS_K3_R8=0 keeps the 4-wave kernel of sdf_kernels.hip,
    read at every call): same stash, same operands, same summation order per accumulator (the sdf tile's two k-steps of the seed
    are added in front of the other sixteen instead of behind them) -- every plane the weight-gradient GEMM reads (qbar, adj_1..8,
    zbar_0..8, zsdf) agrees to a step of its format, and the kernel is bit-reproducible.  64- and 128-sample workgroups
    (FNEUS_R8_NH); ragged sizes as in the reverse sweep's test.  (FNEUS_BWD_XHI=0: the chains on hi + lo activations, the arithmetic
    of the 4-wave kernel; the bf16-activation chains of gradient precision 1 / 2 have their own test below.)"""
    from fneus import ops, synth, pp
    monkeypatch.setenv("FNEUS_R8_NH", str(nh))
    monkeypatch.setenv("FNEUS_BWD_XHI", "0")
    net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(22).items()})
    net.pack()
    g = torch.Generator(device=DEV).manual_seed(7)
    x = (torch.rand(n, 3, device=DEV, generator=g) * 2 - 1).contiguous()
    ds = torch.randn(n, device=DEV, generator=g)
    df = torch.randn(n, 256, device=DEV, generator=g) * 0.1
    dn = torch.randn(n, 3, device=DEV, generator=g)
    st = ops.SdfStash(n, DEV, prec, True, gprec)
    ops.sdf_fwd_grad(net.blob, n, prec, st, True, pts=x)

    def run(r8):
        monkeypatch.setenv("FNEUS_K3_R8", str(r8))
        b = ops.SdfBwdBufs(n, DEV, prec, gprec)
        ops.sdf_bwd(net.blob, n, prec, st, b, ds, df, dn, pts=x)
        torch.cuda.synchronize()
        return b

    b0, b1, b2 = run(0), run(1), run(1)
    for name in ("qbar", "adj", "zbar", "zsdf"):
        t0, t1, t2 = getattr(b0, name), getattr(b1, name), getattr(b2, name)
        assert torch.equal(t1, t2), name
        assert torch.isfinite(t1.float()).all(), name
    assert torch.equal(b1.qbar, b0.qbar) and torch.equal(b1.zsdf, b0.zsdf)
    tol = 8e-3 if gprec == 1 else (2e-4 if prec == 3 else 8e-3)
    if prec == 1:
        tol = 3e-2          # bf16 chain (not a parity mode): a flipped rounding propagates through the layers
    for name, slots in (("adj", 8), ("zbar", 9)):
        t0, t1 = getattr(b0, name), getattr(b1, name)
        for l in range(slots):
            v0, v1 = pp.value(t0[:, l], n), pp.value(t1[:, l], n)
            assert (v1 - v0).abs().max().item() <= tol * max(v0.abs().max().item(), 1e-9), (name, l)
        if t1.shape[2] > b1.tiles:                          # an allocated tile without samples stays zero
            assert float(t1[:, :, b1.tiles:].float().abs().max()) == 0.0


@pytest.mark.parametrize("nh", [4, 2])
@pytest.mark.parametrize("gprec,prec,n", [(1, 3, 40003), (2, 3, 40003), (3, 3, 40003), (1, 1, 40003), (2, 3, 65536), (1, 3, 40067)])
def test_colour_backward_r8_equals_the_4_wave_kernel(monkeypatch, gprec, prec, n, nh):
    """the colour network's backward on resident-weight 8-wave workgroups (csrc/color_r8_kernels.hip, round 6; FNEUS_COL_BWD_R8=0 keeps
    the 4-wave kernel of color_kernels.hip, read at every call): same masks, same operands, same summation order per accumulator --
    d_feat and d_normal agree to the rounding of a lo fragment's last bit, every plane the weight-gradient GEMM reads (zbar_0..3,
    zout) to a step of its format, and the kernel is bit-reproducible.  64- and 128-sample workgroups (FNEUS_R8_NH); ragged sizes.
    (FNEUS_COLB_XHI=0: hi + lo activations inside the chain, the 4-wave kernel's arithmetic.)"""
    from fneus import ops, synth, pp
    monkeypatch.setenv("FNEUS_R8_NH", str(nh))
    monkeypatch.setenv("FNEUS_COLB_XHI", "0")
    net = ops.PackedNet("color", DEV).load_state_dict({k: T(v) for k, v in synth.color_state_dict(23).items()})
    net.pack()
    g = torch.Generator(device=DEV).manual_seed(9)
    x = (torch.rand(n, 3, device=DEV, generator=g) * 2 - 1).contiguous()
    d = torch.randn(n, 3, device=DEV, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    nrm = torch.randn(n, 3, device=DEV, generator=g)
    feat = (torch.randn(n, 256, device=DEV, generator=g) * 0.3).contiguous()
    c_rgb = torch.randn(n, 3, device=DEV, generator=g)
    st = ops.ColStash(n, DEV, prec, gprec=gprec)
    rgb = ops.color_fwd(net.blob, n, prec, nrm, feat, st, True, pts=x, dirs=d)

    def run(r8):
        monkeypatch.setenv("FNEUS_COL_BWD_R8", str(r8))
        st.zbar.zero_()
        st.zout.zero_()
        if st.zout_lo is not None:
            st.zout_lo.zero_()
        df, dn = ops.color_bwd(net.blob, n, prec, c_rgb, rgb, st)
        torch.cuda.synchronize()
        return df.clone(), dn.clone(), st.zbar.clone(), st.zout.clone(), None if st.zout_lo is None else st.zout_lo.clone()

    a, b, c = run(0), run(1), run(1)
    for u, v in zip(b, c):
        assert (u is None and v is None) or torch.equal(u, v)
    assert torch.isfinite(b[0]).all() and torch.isfinite(b[1]).all()
    fast = prec == 1
    sc_f, sc_n = a[0].abs().max().item(), a[1].abs().max().item()
    assert (b[0] - a[0]).abs().max().item() <= (3e-2 if fast else 2e-5) * sc_f
    assert (b[1] - a[1]).abs().max().item() <= (3e-2 if fast else 2e-5) * sc_n
    assert torch.equal(b[3], a[3])                                      # zout: formed from d_rgb and rgb alone
    if a[4] is not None:
        assert torch.equal(b[4], a[4])
    tol = 3e-2 if fast else (8e-3 if gprec != 3 else 2e-4)
    for l in range(4):
        v0, v1 = pp.value(a[2][:, l], n), pp.value(b[2][:, l], n)
        assert (v1 - v0).abs().max().item() <= tol * max(v0.abs().max().item(), 1e-9), l
    if b[2].shape[2] > st.tiles:                                        # an allocated tile without samples stays zero
        assert float(b[2][:, :, st.tiles:].float().abs().max()) == 0.0


@pytest.mark.parametrize("gprec,n", [(1, 40003), (2, 65536), (1, 70001), (1, 40067)])
def test_cotangent_chains_on_bf16_activations(monkeypatch, gprec, n):
    """Round 6: with bf16 gradient planes (gradient precision 1 / 2) the two chains of K3 and the colour network's
    backward run on the bf16 VALUES OF THOSE PLANES -- W hi + lo against one bf16 activation fragment, two MFMAs per product, no lo
    fragments in LDS (FNEUS_BWD_XHI / FNEUS_COLB_XHI, default 1; resident-weight kernels = launches of >= 1024 sample tiles).
    Per sample that is one more rounding of 2^-9 per layer, random in sign: every plane stays within 2e-2 of its scale of the
    hi + lo chain's, 128- and 256-sample workgroups are BIT-identical (the same operands per sample in the same order; 64-sample
    ones to a step of the planes' format), the kernels are bit-reproducible, allocated tiles without samples stay zero.  What the weight gradients lose is bounded against the
    reference's gradients in tests/test_hip_render.py (512-ray fixture) and against fp64 in tests/test_hip_backward.py."""
    from fneus import ops, synth, pp
    net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(22).items()})
    net.pack()
    g = torch.Generator(device=DEV).manual_seed(7)
    x = (torch.rand(n, 3, device=DEV, generator=g) * 2 - 1).contiguous()
    ds = torch.randn(n, device=DEV, generator=g)
    df = torch.randn(n, 256, device=DEV, generator=g) * 0.1
    dn = torch.randn(n, 3, device=DEV, generator=g)
    st = ops.SdfStash(n, DEV, 3, True, gprec)
    ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)

    def k3(xhi, nh):
        monkeypatch.setenv("FNEUS_BWD_XHI", str(xhi))
        monkeypatch.setenv("FNEUS_R8_NH", str(nh))
        b = ops.SdfBwdBufs(n, DEV, 3, gprec)
        ops.sdf_bwd(net.blob, n, 3, st, b, ds, df, dn, pts=x)
        torch.cuda.synchronize()
        return b

    ref, b8, b8b, b4, b2 = k3(0, 4), k3(1, 8), k3(1, 8), k3(1, 4), k3(1, 2)
    for name, slots in (("qbar", 0), ("zsdf", 0), ("adj", 8), ("zbar", 9)):
        t = [getattr(b, name) for b in (b8, b8b, b4, b2)]
        assert torch.equal(t[0], t[1]) and torch.equal(t[0], t[2]), name
        assert torch.isfinite(t[0].float()).all(), name
        # (64-sample workgroups keep c_7 in registers across the turn and hipcc contracts s * ubar + c differently there: a value
        # at a rounding boundary lands on the other bf16 neighbour)
        for l in range(max(slots, 1)):
            u0 = pp.value(t[0][:, l] if slots else t[0], n)
            u3 = pp.value(t[3][:, l] if slots else t[3], n)
            assert (u3 - u0).abs().max().item() <= 8e-3 * max(u0.abs().max().item(), 1e-9), (name, l)
        r = getattr(ref, name)
        if slots == 0:
            assert torch.equal(t[0], r), name                       # formed from the cotangents alone
            continue
        for l in range(slots):
            v0, v1 = pp.value(r[:, l], n), pp.value(t[0][:, l], n)
            assert (v1 - v0).abs().max().item() <= 2e-2 * max(v0.abs().max().item(), 1e-9), (name, l)
            assert (v1 - v0).norm().item() <= 8e-3 * max(v0.norm().item(), 1e-9), (name, l)      # observed 2.7e-3 ... 5.0e-3
        if t[0].shape[2] > b8.tiles:
            assert float(t[0][:, :, b8.tiles:].float().abs().max()) == 0.0

    cnet = ops.PackedNet("color", DEV).load_state_dict({k: T(v) for k, v in synth.color_state_dict(23).items()})
    cnet.pack()
    d = torch.randn(n, 3, device=DEV, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    nrm = torch.randn(n, 3, device=DEV, generator=g)
    feat = (torch.randn(n, 256, device=DEV, generator=g) * 0.3).contiguous()
    c_rgb = torch.randn(n, 3, device=DEV, generator=g)
    cs = ops.ColStash(n, DEV, 3, gprec=gprec)
    rgb = ops.color_fwd(cnet.blob, n, 3, nrm, feat, cs, True, pts=x, dirs=d)

    def cb(xhi, nh):
        monkeypatch.setenv("FNEUS_COLB_XHI", str(xhi))
        monkeypatch.setenv("FNEUS_R8_NH", str(nh))
        cs.zbar.zero_()
        cs.zout.zero_()
        if cs.zout_lo is not None:
            cs.zout_lo.zero_()
        dfe, dno = ops.color_bwd(cnet.blob, n, 3, c_rgb, rgb, cs)
        torch.cuda.synchronize()
        return dfe.clone(), dno.clone(), cs.zbar.clone(), cs.zout.clone(), None if cs.zout_lo is None else cs.zout_lo.clone()

    ref, c8, c8b, c4, c2 = cb(0, 4), cb(1, 8), cb(1, 8), cb(1, 4), cb(1, 2)
    for other in (c8b, c4, c2):
        for u, v in zip(c8, other):
            assert (u is None and v is None) or torch.equal(u, v)
    assert torch.isfinite(c8[0]).all() and torch.isfinite(c8[1]).all()
    assert torch.equal(c8[3], ref[3]) and (ref[4] is None or torch.equal(c8[4], ref[4]))          # zout hi (+ lo): from d_rgb and rgb alone
    for i in (0, 1):                                                                                # d_feat, d_normal
        assert (c8[i] - ref[i]).abs().max().item() <= 2e-2 * ref[i].abs().max().item()
        assert (c8[i] - ref[i]).norm().item() <= 8e-3 * ref[i].norm().item()                     # observed 3.8e-3, 3.9e-3
    for l in range(4):
        v0, v1 = pp.value(ref[2][:, l], n), pp.value(c8[2][:, l], n)
        assert (v1 - v0).abs().max().item() <= 2e-2 * max(v0.abs().max().item(), 1e-9), l
    if c8[2].shape[2] > cs.tiles:
        assert float(c8[2][:, :, cs.tiles:].float().abs().max()) == 0.0


@pytest.mark.parametrize("n", [40003, 65536, 100])
def test_colour_output_layer_gradient_with_exact_operands(n):
    """fneus_color_out_dw (gradient precision 2): dW4 = zout^T u_3 and db4 = sum zout from the hi + lo planes of u_3 and zout formed in
    fp32 from d_rgb and rgb, against fp64 on the same planes: fp32 accumulation only (<= 1e-5 of the tensor's largest element)"""
    from fneus import ops, synth, pp
    net = ops.PackedNet("color", DEV).load_state_dict({k: T(v) for k, v in synth.color_state_dict(24).items()})
    g = torch.Generator(device=DEV).manual_seed(13)
    u3 = torch.randn(n, 256, device=DEV, generator=g).clamp_min(0.0)
    st = ops.ColStash(n, DEV, 3, gprec=2)
    planes = pp.pack(u3, 16, 2)                                          # [2, tiles, 16, 64, 8]
    st.u[0, 3].copy_(planes[0])
    st.u3_lo.copy_(planes[1])
    rgb = torch.rand(n, 3, device=DEV, generator=g)
    d_rgb = torch.randn(n, 3, device=DEV, generator=g)
    grad = torch.zeros(net.n_params, dtype=torch.float32, device=DEV)
    ops.color_out_dw(net, st, d_rgb, rgb, grad, n)
    dWs, dbs = net.split_flat(grad)
    uval = pp.value(planes, n).double()                                  # what the planes hold (17 significant bits)
    z = (d_rgb * rgb * (1.0 - rgb)).double()
    ref_W, ref_b = z.t() @ uval, z.sum(0)
    assert (dWs[4].double() - ref_W).abs().max().item() <= 1e-5 * ref_W.abs().max().item()
    assert (dbs[4].double() - ref_b).abs().max().item() <= 1e-5 * max(ref_b.abs().max().item(), 1.0)
    for l in range(4):
        assert float(dWs[l].abs().max()) == 0.0 and float(dbs[l].abs().max()) == 0.0


@pytest.mark.parametrize("n", [65536, 40003, 32768 + 64, 1000])
@pytest.mark.parametrize("gprec", [1, 3])
def test_feature_planes_are_complete_in_every_training_mode(n, gprec):
    """round 6: the SDF stash's feature planes are the colour network's INPUT -- hi + lo in every training mode of the parity
    arithmetic, every sample tile written by every form of K2 (the two-launch form incl. the tail of each workgroup's last unit, the
    fused kernel of small launches), with or without the fp32 rows.  The planes are poisoned with NaN before the launch."""
    from fneus import ops, synth, pp
    net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(25).items()})
    net.pack()
    x = (torch.rand(n, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)) * 2 - 1).contiguous()
    st = ops.SdfStash(n, DEV, 3, True, gprec)
    assert st.feat.shape[0] == 2
    st.feat.fill_(float("nan"))
    sdf, feat, nrm = ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
    torch.cuda.synchronize()
    planes = st.feat[:, :st.tiles].float()
    assert torch.isfinite(planes).all()
    val = pp.value(st.feat[:, :st.tiles], n)
    assert (val - feat).abs().max().item() <= 2.0 ** -15 * feat.abs().max().item()
    if ops.feat_planes_ok(n, 3, True):
        st2 = ops.SdfStash(n, DEV, 3, True, gprec)
        st2.feat.fill_(float("nan"))
        sdf2, ph, nrm2 = ops.sdf_fwd_grad(net.blob, n, 3, st2, True, pts=x, feat_rows=False)
        torch.cuda.synchronize()
        assert ph.planes_of is st2
        assert torch.equal(st2.feat[:, :st2.tiles], st.feat[:, :st.tiles]) and torch.equal(sdf2, sdf) and torch.equal(nrm2, nrm)


def test_k1_on_marked_rays_only_equals_k1_there_and_fills_the_rest():
    """fneus_sdf_fwd_rays (the stage-2 march of the fixed-shape step): marked rays get the values of the plain launch bit for bit,
    the samples of the others the fill value; nothing marked, everything marked and a ragged mix"""
    import numpy as np
    from fneus import ops, synth
    dev = torch.device("cuda:0")
    net = ops.PackedNet("sdf", dev).load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.sdf_state_dict(7).items()})
    net.pack()
    R, m = 300, 256
    g = torch.Generator().manual_seed(0)
    o = (torch.rand(R, 3, generator=g) - 0.5).to(dev).contiguous()
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev).contiguous()
    t = torch.rand(R * m, generator=g).to(dev).contiguous()
    full = ops.sdf_fwd(net.blob, R * m, 3, rays_o=o, rays_d=d, t=t, m=m)
    for frac in (0.0, 1.0, 0.37):
        mask = (torch.rand(R, generator=g) < frac).to(dev)
        got = ops.sdf_fwd(net.blob, R * m, 3, rays_o=o, rays_d=d, t=t, m=m, ray_mask=mask, fill=2.5)
        want = torch.where(mask[:, None], full.reshape(R, m), torch.full_like(full.reshape(R, m), 2.5))
        assert torch.equal(got.reshape(R, m), want), frac


@pytest.mark.parametrize("n", [65536, 40003])
def test_feature_cotangent_fragments_equal_the_rounded_rows(monkeypatch, n):
    """fneus_color_bwd with d_feat == NULL (FneusColStash.dfeat_hi) stores bf16(d_feat) as fragments -- bit for bit the rows' rounding,
    zeros behind a ragged end; fneus_sdf_bwd with d_feat == NULL seeds its descending chain from them: every plane it writes equals the
    launch on the rows; fneus_surface_scatter_plane adds the heads' rows into the fragments (fp32 sum, rounded once more)."""
    from fneus import ops, synth, pp
    assert ops.dfeat_plane_ok(n, 3, 2)
    g = torch.Generator(device=DEV).manual_seed(19)
    x = (torch.rand(n, 3, device=DEV, generator=g) * 2 - 1).contiguous()
    cnet = ops.PackedNet("color", DEV).load_state_dict({k: T(v) for k, v in synth.color_state_dict(23).items()})
    cnet.pack()
    d = torch.randn(n, 3, device=DEV, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    nrm = torch.randn(n, 3, device=DEV, generator=g)
    feat = (torch.randn(n, 256, device=DEV, generator=g) * 0.3).contiguous()
    c_rgb = torch.randn(n, 3, device=DEV, generator=g)
    cs = ops.ColStash(n, DEV, 3, gprec=2)
    rgb = ops.color_fwd(cnet.blob, n, 3, nrm, feat, cs, True, pts=x, dirs=d)
    rows, dn_rows = ops.color_bwd(cnet.blob, n, 3, c_rgb, rgb, cs)
    zb_rows = cs.zbar.clone()
    T_al = 2 * ((n + 63) // 64)
    plane = torch.full((T_al, 16, 64, 8), float("nan"), dtype=torch.bfloat16, device=DEV)
    ph, dn_plane = ops.color_bwd(cnet.blob, n, 3, c_rgb, rgb, cs, dfeat_plane=plane)
    torch.cuda.synchronize()
    assert ph.plane_of is plane and torch.equal(dn_plane, dn_rows) and torch.equal(cs.zbar, zb_rows)
    got = pp.unpack(plane)                                           # [T_al * 32, 256]
    assert torch.equal(got[:n], rows.bfloat16())
    assert float(got[n:(n + 31) // 32 * 32].float().abs().max()) == 0.0 if n % 32 else True
    # ---- K3 on the fragments against K3 on the rows
    net = ops.PackedNet("sdf", DEV).load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(22).items()})
    net.pack()
    ds, dnn = torch.randn(n, device=DEV, generator=g), torch.randn(n, 3, device=DEV, generator=g)
    st = ops.SdfStash(n, DEV, 3, True, 2)
    ops.sdf_fwd_grad(net.blob, n, 3, st, True, pts=x)
    b0, b1 = ops.SdfBwdBufs(n, DEV, 3, 2), ops.SdfBwdBufs(n, DEV, 3, 2)
    ops.sdf_bwd(net.blob, n, 3, st, b0, ds, rows, dnn, pts=x)
    b1.zbar[0, 8].copy_(plane)
    ops.sdf_bwd(net.blob, n, 3, st, b1, ds, None, dnn, pts=x)
    torch.cuda.synchronize()
    for name in ("qbar", "adj", "zbar", "zsdf"):
        assert torch.equal(getattr(b0, name), getattr(b1, name)), name
    # ---- the heads' rows into the fragments
    R = 1024
    sel = torch.randperm(n, device=DEV, generator=g)[:R].to(torch.int32).contiguous()
    dfh = (torch.randn(2, R, 256, device=DEV, generator=g) * rows.abs().mean()).contiguous()
    dnh = torch.randn(2, R, 3, device=DEV, generator=g).contiguous()
    rows2, dn2 = rows.clone(), dnn.clone()
    ops.surface_scatter(sel, dfh, dnh, rows2, dn2)
    p2, dn3 = plane.clone(), dnn.clone()
    ops.surface_scatter_plane(sel, dfh, dnh, p2, n, dn3)
    torch.cuda.synchronize()
    assert torch.equal(dn3, dn2)
    got2 = pp.unpack(p2)[:n].float()
    untouched = torch.ones(n, dtype=torch.bool, device=DEV)
    untouched[sel.long()] = False
    assert torch.equal(got2[untouched], got[:n].float()[untouched])
    want = ((rows.bfloat16().float()[sel.long()] + dfh[0]) + dfh[1]).bfloat16().float()          # (the kernel's order of the fp32 sum)
    assert torch.equal(got2[sel.long()], want)
    # the kernels that cannot take the fragments refuse loudly
    monkeypatch.setenv("FNEUS_BWD_XHI", "0")
    with pytest.raises(RuntimeError):
        ops.sdf_bwd(net.blob, n, 3, st, b1, ds, None, dnn, pts=x)
