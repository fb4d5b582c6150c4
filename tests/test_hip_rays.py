"""GPU parity of the per-ray kernels (sampler pieces, NeuS alpha/compositing fwd+bwd) vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_upsample_and_merge_golden(golden_dir):
    """reference fixtures (tests/golden/units.npz): up_sample new z, and cat_z_vals-style merge"""
    from fneus import ops
    from oracle import ref_torch as R
    g = dict(np.load(os.path.join(golden_dir, "units.npz")))
    ro, rd, z, s = (T(g[k]).to(DEV).contiguous() for k in ("ups_rays_o", "ups_rays_d", "ups_z", "ups_sdf"))
    for inv_s in (64, 512):
        out = ops.upsample(ro, rd, z, s, 8, inv_s).cpu()
        err = (out - T(g[f"ups_new_z_{inv_s}"])).abs()
        assert err.max().item() <= 5e-4 and err.median().item() <= 2e-6, (inv_s, err.max(), err.median())
        # merge: compare with torch.sort of the concatenation
        new_z = T(g[f"ups_new_z_{inv_s}"]).to(DEV).contiguous()
        new_s = torch.sin(new_z * 3.0).contiguous()
        zm, sm = ops.merge(z, s, new_z, new_s)
        zc, idx = torch.sort(torch.cat([z, new_z], -1), dim=-1, stable=True)
        sc = torch.gather(torch.cat([s, new_s], -1), 1, idx)
        assert torch.equal(zm, zc) and torch.equal(sm, sc)
        zm2, none = ops.merge(z, None, new_z, None)
        assert torch.equal(zm2, zc) and none is None


@pytest.mark.parametrize("m,k", [(64, 16), (80, 16), (112, 16), (16, 4), (33, 7)])
def test_upsample_random(m, k):
    from fneus import ops, synth
    from oracle import ref_torch as R
    rs = np.random.RandomState(m)
    B = 50
    data = T(synth.ray_batch(B, seed=m, n_miss=3))
    ro, rd = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    near, far = R.near_far_from_sphere(ro, rd)
    z = near + (far - near) * torch.linspace(0, 1, m)[None, :]
    pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
    sdf = (pts.norm(dim=-1) - 0.5 + 0.05 * torch.sin(7 * pts[..., 0])).float()
    for inv_s in (64.0, 256.0):
        ref = R.up_sample(ro, rd, z, sdf, k, inv_s)
        out = ops.upsample(ro.to(DEV), rd.to(DEV), z.to(DEV).contiguous(), sdf.to(DEV).contiguous(), k, inv_s).cpu()
        err = (out - ref).abs()
        assert err.max().item() <= 1e-3 and err.median().item() <= 5e-6, (err.max(), err.median())


@pytest.mark.parametrize("m,k,last", [(64, 16, False), (80, 16, False), (96, 16, True), (40, 8, True), (33, 7, False)])
def test_fused_merge_upsample_is_bit_identical_to_the_separate_kernels(m, k, last):
    """fneus_merge_upsample = fneus_merge -> fneus_upsample (-> fneus_merge without sdf on the last step)"""
    from fneus import ops, synth
    B = 97
    data = T(synth.ray_batch(B, seed=m + k, n_miss=5)).to(DEV)
    ro, rd = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    g = torch.Generator().manual_seed(m)
    z = torch.sort(torch.rand(B, m, generator=g) * 2.0 + 0.5, dim=-1)[0].to(DEV).contiguous()
    new_z = torch.sort(torch.rand(B, k, generator=g) * 2.0 + 0.5, dim=-1)[0].to(DEV).contiguous()
    new_z[:, 0] = z[:, 3]                                  # ties keep the old sample first
    pts = lambda t: ro[:, None, :] + rd[:, None, :] * t[..., None]
    f = lambda t: (pts(t).norm(dim=-1) - 0.5 + 0.05 * torch.sin(7 * pts(t)[..., 0])).float().contiguous()
    s, new_s = f(z), f(new_z)
    inv_s = 128.0
    z1, s1 = ops.merge(z, s, new_z, new_s)
    nz = ops.upsample(ro, rd, z1, s1, k, inv_s)
    zf = ops.merge(z1, None, nz, None)[0] if last else None
    z2, s2, nz2, zf2 = ops.merge_upsample(ro, rd, z, s, new_z, new_s, inv_s, k, last)
    assert torch.equal(z1, z2) and torch.equal(s1, s2) and torch.equal(nz, nz2)
    assert (zf2 is None) == (not last) and (not last or torch.equal(zf, zf2))
    # with a section length given, the last launch also writes fneus_sections of its result
    sd = 2.0 / 64
    z3, s3, nz3, zf3, dists3, mid3 = ops.merge_upsample(ro, rd, z, s, new_z, new_s, inv_s, k, last, sample_dist=sd)
    assert torch.equal(z3, z2) and torch.equal(s3, s2) and torch.equal(nz3, nz2)
    if last:
        dists, mid = ops.sections(zf, sd)
        assert torch.equal(zf3, zf) and torch.equal(dists3, dists) and torch.equal(mid3, mid)
    else:
        assert zf3 is None and dists3 is None and mid3 is None
    # round 5: the fused kernel ranks an entry by its place in its own run + a search of the other run.  Only the OLD run has to
    # be ascending for that; new depths in any order (and with ties among themselves) still come out as the stable sort
    perm = torch.randperm(k, generator=g).to(DEV)
    new_z_p, new_s_p = new_z[:, perm].contiguous(), new_s[:, perm].contiguous()
    new_z_p[:, 1] = new_z_p[:, 2]
    new_s_p[:, 1] = 0.125
    z4, s4 = ops.merge(z, s, new_z_p, new_s_p)
    z5, s5, _, _ = ops.merge_upsample(ro, rd, z, s, new_z_p, new_s_p, inv_s, k, last)
    assert torch.equal(z4, z5) and torch.equal(s4, s5)


def test_fused_sampler_in_render_is_bit_identical():
    """NeuSRenderer._hierarchical_z: 7 launches (fused) against the reference's call sequence of 11"""
    import models.renderer as MR
    from fneus import synth
    from fneus.trainer import Stage1Trainer
    tr = Stage1Trainer(torch.device(DEV), seed=3, use_graph=False)
    data = T(synth.ray_batch(300, seed=12, n_miss=20)).to(DEV)
    ro, rd = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    z0 = fops().ray_setup(ro, rd, tr.renderer.n_samples)
    tr.renderer.sdf_network.refresh()
    old = MR.SAMPLER_FUSED
    try:
        MR.SAMPLER_FUSED = True
        a = tr.renderer._hierarchical_z(ro, rd, z0)
        MR.SAMPLER_FUSED = False
        b = tr.renderer._hierarchical_z(ro, rd, z0)
    finally:
        MR.SAMPLER_FUSED = old
    assert a.shape == (300, 128) and torch.equal(a, b)


@pytest.mark.parametrize("B,k,last", [(512, 16, False), (512, 16, True), (301, 16, True), (64, 32, False), (7, 16, True)])
def test_sampler_step_inside_the_k1_launch_is_bit_identical(B, k, last):
    """round 6: fneus_sdf_fwd_merge_upsample = fneus_sdf_fwd on the step's new depths followed by fneus_merge_upsample, in ONE launch (a
    32-sample tile of the evaluation is 32 / k whole rays; the workgroup that evaluated it merges them, the new sdf values from LDS):
    every output bit for bit, ragged ray counts (a last tile with one ray), both tile shapes, with and without the last step's sections"""
    from fneus import ops, synth
    net = ops.PackedNet("sdf", DEV).load_state_dict({k_: T(v) for k_, v in synth.sdf_state_dict(26).items()})
    net.pack()
    data = T(synth.ray_batch(B, seed=B + k, n_miss=max(1, B // 20))).to(DEV)
    ro, rd = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    m = 80
    g = torch.Generator().manual_seed(B)
    z = torch.sort(torch.rand(B, m, generator=g) * 2.0 + 0.5, dim=-1)[0].to(DEV).contiguous()
    new_z = torch.sort(torch.rand(B, k, generator=g) * 2.0 + 0.5, dim=-1)[0].to(DEV).contiguous()
    s = ops.sdf_fwd(net.blob, B * m, 3, rays_o=ro, rays_d=rd, t=z.reshape(-1), m=m).reshape(B, m)
    new_s = ops.sdf_fwd(net.blob, B * k, 3, rays_o=ro, rays_d=rd, t=new_z.reshape(-1), m=k).reshape(B, k).contiguous()
    sd = 2.0 / 64
    ref = ops.merge_upsample(ro, rd, z, s, new_z, new_s, 256.0, 16, last, sample_dist=sd)
    got = ops.sdf_merge_upsample(net.blob, 3, ro, rd, z, s, new_z, 256.0, 16, last, sample_dist=sd, want_s_new=True)
    assert got is not None
    assert torch.equal(got[-1], new_s)
    for a, b, name in zip(got[:-1], ref, ("z_out", "s_out", "z_next", "z_final", "dists", "mid_z")):
        assert (a is None) == (b is None), name
        if a is not None:
            assert torch.equal(a, b), name
    # shapes the launch does not take: the caller falls back
    assert ops.sdf_merge_upsample(net.blob, 3, ro, rd, z, s, new_z[:, :8].contiguous(), 256.0, 16, last, sample_dist=sd) is None


@pytest.mark.parametrize("B,k,steps", [(512, 16, 4), (37, 16, 4), (64, 32, 3), (512, 16, 3)])
def test_all_sampler_steps_in_one_launch_are_bit_identical(monkeypatch, B, k, steps):
    """round 6: fneus_sdf_fwd_merge_upsample_steps runs every remaining step of the hierarchical sampler in ONE launch (a workgroup keeps
    its 32 / k rays through the steps and evaluates the depths its own merge has drawn): final depths and sections bit for bit those of
    the step-by-step launches, ragged ray counts, both tile shapes."""
    from fneus import ops, synth
    net = ops.PackedNet("sdf", DEV).load_state_dict({k_: T(v) for k_, v in synth.sdf_state_dict(27).items()})
    net.pack()
    data = T(synth.ray_batch(B, seed=B + k, n_miss=max(1, B // 20))).to(DEV)
    ro, rd = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    m = 64
    z = ops.ray_setup(ro, rd, m)
    s = ops.sdf_fwd(net.blob, B * m, 3, rays_o=ro, rays_d=rd, t=z.reshape(-1), m=m).reshape(B, m).contiguous()
    new_z = ops.upsample(ro, rd, z, s, k, 64.0)
    sd = 2.0 / 64
    inv = [float(64 * 2 ** i) for i in range(1, steps)]
    got = ops.sdf_merge_upsample_steps(net.blob, 3, ro, rd, z, s, new_z, inv, k, sd)
    assert got is not None
    zc, sc, nz = z, s, new_z
    for j, inv_s in enumerate(inv):
        ref = ops.sdf_merge_upsample(net.blob, 3, ro, rd, zc, sc, nz.contiguous(), inv_s, k, j + 1 == len(inv), sample_dist=sd)
        assert ref is not None
        zc, sc, nz = ref[0], ref[1], ref[2]
    torch.cuda.synchronize()
    for a, b, name in zip(got, ref[3:6], ("z_final", "dists", "mid_z")):
        assert torch.equal(a, b), name
    assert got[0].shape == (B, m + k * steps)
    # one step alone is not this entry point's case
    assert ops.sdf_merge_upsample_steps(net.blob, 3, ro, rd, z, s, new_z, inv[:1], k, sd) is None


def fops():
    from fneus import ops
    return ops


def test_sections():
    from fneus import ops
    z = torch.sort(torch.rand(7, 40), dim=-1)[0]
    dists, mid = ops.sections(z.to(DEV), 2.0 / 16)
    d_ref = torch.cat([z[:, 1:] - z[:, :-1], torch.full((7, 1), 2.0 / 16)], -1)
    assert torch.equal(dists.cpu(), d_ref) and torch.equal(mid.cpu(), z + d_ref * 0.5)


def make_fields(B, n, seed):
    from fneus import synth
    from oracle import ref_torch as R
    rs = np.random.RandomState(seed)
    data = T(synth.ray_batch(B, seed=seed, n_miss=2))
    ro, rd = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    near, far = R.near_far_from_sphere(ro, rd)
    z = torch.sort(near + (far - near) * T(rs.uniform(0, 1, size=(B, n)).astype(np.float32)), dim=-1)[0]
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((B, 1), 2.0 / 64)], -1)
    mid_z = z + dists * 0.5
    pts = ro[:, None, :] + rd[:, None, :] * mid_z[..., None]
    sdf = (pts.norm(dim=-1) - 0.6 + 0.03 * torch.sin(9 * pts[..., 1])).reshape(-1)
    sdf[: n] = sdf[:n].abs() + 0.01          # ray 0: never negative
    normal = pts.reshape(-1, 3) / pts.reshape(-1, 3).norm(dim=-1, keepdim=True) * T(rs.uniform(0.7, 1.3, size=(B * n, 1)).astype(np.float32))
    normal = normal + 0.1 * T(rs.standard_normal((B * n, 3)).astype(np.float32))
    rgb = T(rs.uniform(0, 1, size=(B * n, 3)).astype(np.float32))
    return ro, rd, mid_z.contiguous(), dists.contiguous(), sdf.contiguous(), normal.contiguous(), rgb


@pytest.mark.parametrize("n,car,inv_s", [(128, 1.0, 20.0), (32, 0.3, 80.0), (160, 0.0, 300.0), (37, 1.0, 50.0)])
def test_composite_fwd_bwd(n, car, inv_s):
    from fneus import ops
    from oracle import ref_torch as R
    B = 24
    ro, rd, mid_z, dists, sdf, normal, rgb = make_fields(B, n, seed=n)
    rs = np.random.RandomState(n + 1)
    # oracle in fp64 with autograd
    sdf64 = sdf.double().requires_grad_(True)
    nrm64 = normal.double().requires_grad_(True)
    rgb64 = rgb.double().requires_grad_(True)
    s64 = torch.tensor(inv_s, dtype=torch.float64, requires_grad=True)
    ref = R.composite_from_fields(ro.double(), rd.double(), mid_z.double(), dists.double(), sdf64, nrm64, rgb64, s64, car)
    c_col = T(rs.standard_normal((B, 3)))
    c_ws = T(rs.standard_normal(B))
    c_w = T(rs.standard_normal((B, n)) * 0.1)
    c_pair = T(rs.standard_normal((B, 2)))
    c_eik = T(rs.standard_normal(B) * 0.01)
    L = (ref["color"] * c_col).sum() + (ref["wsum"] * c_ws).sum() + (ref["weights"] * c_w).sum() + \
        (ref["wpair"] * c_pair).sum() + (ref["eik_num"] * c_eik).sum()
    L.backward()
    # HIP
    d = lambda t: t.float().to(DEV).contiguous()
    inv_s_dev = torch.tensor([inv_s], dtype=torch.float32, device=DEV)
    out = ops.composite_fwd(d(ro), d(rd), d(mid_z), d(dists), d(sdf), d(normal), d(rgb), inv_s_dev, car)
    assert torch.equal(out["sdf_mask"].cpu().bool(), ref["sdf_mask"])
    m = ref["sdf_mask"]
    assert torch.equal(out["min_idx"].cpu().long()[m], ref["min_idx"][m])
    assert torch.equal(out["inside"].cpu().double(), ref["inside"])
    for k, rk in (("weights", "weights"), ("color", "color"), ("wsum", "wsum"), ("wmax", "wmax"), ("cdf", "cdf"),
                  ("wpair", "wpair")):
        e = (out[k].cpu().double() - ref[rk].detach()).abs().max().item()
        assert e <= 2e-6, (k, e)
    assert (out["eik"][0].cpu().double() - ref["eik_num"].detach()).abs().max().item() <= 1e-5
    assert torch.equal(out["eik"][1].cpu().double(), ref["eik_den"])
    d_sdf, d_nrm, d_rgb, d_inv, _, _ = ops.composite_bwd(d(ro), d(rd), d(mid_z), d(dists), d(sdf), d(normal), d(rgb),
                                                         inv_s_dev, car, out["min_idx"], out["sdf_mask"], d(c_col),
                                                         d(c_ws), d(c_w), d(c_pair), d(c_eik))

    def rel(a, b):
        return ((a.cpu().double() - b).abs().max() / (b.abs().max() + 1e-30)).item()

    e1, e2, e3 = rel(d_sdf, sdf64.grad), rel(d_nrm, nrm64.grad), rel(d_rgb, rgb64.grad)
    e4 = abs(d_inv.sum().item() - s64.grad.item()) / (abs(s64.grad.item()) + 1e-30)
    print(f"composite_bwd n={n}: d_sdf {e1:.2e} d_normal {e2:.2e} d_rgb {e3:.2e} d_inv_s {e4:.2e}")
    assert e1 <= 2e-4 and e2 <= 2e-4 and e3 <= 1e-5 and e4 <= 2e-4


@pytest.mark.parametrize("n,n_out", [(32, 8), (128, 32)])
def test_composite_with_background(n, n_out):
    """womask blend (renderer.py:350-356): composite of the SDF branch with background alpha / colours, fwd + bwd,
    against torch autograd of the oracle's formulas in fp64"""
    from fneus import ops
    from oracle import ref_torch as R
    B = 16
    ro, rd, mid_z, dists, sdf, normal, rgb = make_fields(B, n, seed=100 + n)
    rs = np.random.RandomState(n)
    nt = n + n_out
    bga = T(rs.uniform(0.0, 0.3, size=(B, nt)).astype(np.float32))
    bgc = T(rs.uniform(0, 1, size=(B, nt, 3)).astype(np.float32))
    inv_s, car = 40.0, 0.5
    sdf64, nrm64, rgb64 = (t.double().requires_grad_(True) for t in (sdf, normal, rgb))
    bga64, bgc64 = bga.double().requires_grad_(True), bgc.double().requires_grad_(True)
    s64 = torch.tensor(inv_s, dtype=torch.float64, requires_grad=True)
    ref = R.composite_from_fields(ro.double(), rd.double(), mid_z.double(), dists.double(), sdf64, nrm64, rgb64, s64, car)
    ins = ref["inside"]
    alpha = torch.cat([ref["alpha"] * ins + bga64[:, :n] * (1.0 - ins), bga64[:, n:]], -1)
    col = torch.cat([rgb64.reshape(B, n, 3) * ins[..., None] + bgc64[:, :n] * (1.0 - ins)[..., None], bgc64[:, n:]], 1)
    w = alpha * R.exclusive_transmittance(alpha)
    color = (col * w[..., None]).sum(1)
    c_col, c_w, c_pair = T(rs.standard_normal((B, 3))), T(rs.standard_normal((B, nt)) * 0.1), T(rs.standard_normal((B, 2)))
    L = (color * c_col).sum() + (w * c_w).sum() + (ref["wpair"] * c_pair).sum() + w.sum(-1).sum() * 0.3
    L.backward()
    d = lambda t: t.float().to(DEV).contiguous()
    inv_s_dev = torch.tensor([inv_s], dtype=torch.float32, device=DEV)
    out = ops.composite_fwd(d(ro), d(rd), d(mid_z), d(dists), d(sdf), d(normal), d(rgb), inv_s_dev, car, d(bga), d(bgc))
    assert (out["weights"].cpu().double() - w.detach()).abs().max().item() <= 2e-6
    assert (out["color"].cpu().double() - color.detach()).abs().max().item() <= 2e-6
    assert (out["wpair"].cpu().double() - ref["wpair"].detach()).abs().max().item() <= 2e-6
    g = ops.composite_bwd(d(ro), d(rd), d(mid_z), d(dists), d(sdf), d(normal), d(rgb), inv_s_dev, car, out["min_idx"],
                          out["sdf_mask"], d(c_col), torch.full((B,), 0.3, device=DEV), d(c_w), d(c_pair),
                          torch.zeros(B, device=DEV), d(bga), d(bgc))
    rel = lambda a, b: ((a.cpu().double() - b).abs().max() / (b.abs().max() + 1e-30)).item()
    errs = [rel(g[0], sdf64.grad), rel(g[1], nrm64.grad), rel(g[2], rgb64.grad), rel(g[4], bga64.grad), rel(g[5], bgc64.grad)]
    e_s = abs(g[3].sum().item() - s64.grad.item()) / (abs(s64.grad.item()) + 1e-30)
    print("composite+bg rel errs", ["%.1e" % e for e in errs], "inv_s %.1e" % e_s)
    assert max(errs) <= 2e-4 and e_s <= 2e-4


@pytest.mark.parametrize("n,perturb", [(64, True), (64, False), (17, True)])
def test_ray_setup_matches_torch_formulation(n, perturb):
    """near_far_from_sphere (dataset.py:186-192) + the coarse depths and jitter of renderer.py:393-409 in one launch"""
    from fneus import ops
    from oracle import ref_torch as R
    rs = np.random.RandomState(5)
    B = 300
    ro = torch.from_numpy(rs.uniform(-2, 2, size=(B, 3)).astype(np.float32)).to(DEV)
    rd = torch.from_numpy(rs.standard_normal((B, 3)).astype(np.float32)).to(DEV)
    rd = rd / rd.norm(dim=-1, keepdim=True)
    t_rand = torch.from_numpy(rs.uniform(size=(B, 1)).astype(np.float32)).to(DEV) if perturb else None
    near, far = R.near_far_from_sphere(ro, rd)
    z = near + (far - near) * torch.linspace(0.0, 1.0, n, device=DEV)[None, :]
    if perturb:
        z = z + (t_rand - 0.5) * 2.0 / n
    got = ops.ray_setup(ro, rd, n, t_rand=t_rand)
    assert (got - z).abs().max().item() <= 5e-7
    got2 = ops.ray_setup(ro, rd, n, near=near.reshape(-1).contiguous(), far=far.reshape(-1).contiguous(), t_rand=t_rand)
    assert (got2 - z).abs().max().item() <= 2.5e-7      # given near / far: the same roundings as torch


SAMPLER_CASES = ["render_wmask_b16_n16", "render_wmask_b8_n64", "render_womask_b16_n16_o8", "render_wmask_b256_n32",
                 "render_wmask_b64_n64", "render_wmask_b512_n64"]


@pytest.mark.parametrize("name", SAMPLER_CASES)
def test_upsample_bins_and_depths_vs_reference_trace(golden_dir, name):
    """fneus_upsample against the reference's own sampler trace (renderer.py:43-77, 152-189): every up-sampling step is
    fed the reference's inputs of that step, and for every new depth we compare (a) the cdf bin it was drawn from with
    the index torch.searchsorted returned in the reference, (b) the depth itself.  This separates "the kernel picks
    another bin" from "the inverse cdf is ill conditioned":
      * a different bin is accepted only where the sample's u sits within rounding (1e-6) of the cdf value that
        separates the two bins -- there both choices give the same depth up to the bin edge;
      * inside the same bin, |dz| <= 4e-6 * max(1, z width / cdf mass of the bin) + 2e-6: the lerp amplifies the fp32
        rounding of the cdf (different summation order) by that conditioning factor -- it, not the kernel, is what
        exceeds 1e-4 in flat bins.
    Reported: the fraction of new depths within 1e-4 of the reference's."""
    from fneus import ops
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    s = int(g["ray_stride"]) if "ray_stride" in g else 1
    data = T(g["data"])[::s]
    ro, rd = data[:, :3].contiguous().to(DEV), data[:, 3:6].contiguous().to(DEV)
    k = int(g["n_importance"]) // 4
    u = torch.linspace(0.5 / k, 1.0 - 0.5 / k, k)[None, :]
    tot = within = flips = 0
    for i in range(4):
        z_in, sdf_in = T(g[f"trace/z_in_{i}"]), T(g[f"trace/sdf_in_{i}"])
        cdf, ref_bin, ref_z = T(g[f"trace/cdf_{i}"]), T(g[f"trace/bin_{i}"].astype(np.int64)), T(g[f"trace/new_z_{i}"])
        out = ops.upsample(ro, rd, z_in.to(DEV).contiguous(), sdf_in.to(DEV).contiguous(), k, float(64 * 2 ** i)).cpu()
        m = z_in.shape[1]
        hip_bin = (torch.searchsorted(z_in.contiguous(), out.contiguous(), right=True) - 1).clamp(0, m - 2)
        same = hip_bin == ref_bin
        # (a) bin flips: u within rounding of the separating cdf value
        edge = torch.gather(cdf, 1, torch.maximum(hip_bin, ref_bin))
        explained = (u.expand_as(edge) - edge).abs() <= 1e-6
        # a zero-width z bin (duplicate depths) makes the z-space bin ambiguous: accept when the depths coincide
        coincide = (out - ref_z).abs() <= 1e-6
        bad = ~same & ~explained & ~coincide
        assert not bool(bad.any()), (name, i, int(bad.sum()), hip_bin[bad][:4], ref_bin[bad][:4])
        # (b) depths inside the same bin
        nxt = (ref_bin + 1).clamp(max=m - 1)
        zw = torch.gather(z_in, 1, nxt) - torch.gather(z_in, 1, ref_bin)
        cw = (torch.gather(cdf, 1, nxt) - torch.gather(cdf, 1, ref_bin)).clamp(min=1e-5)
        err = (out - ref_z).abs()
        lim = 4e-6 * torch.clamp(zw / cw, min=1.0) + 2e-6      # cdf rounding (a few 1e-7 per scan step) x conditioning
        assert bool((err[same] <= lim[same]).all()), (name, i, (err - lim)[same].max().item())
        tot += err.numel()
        within += int((err <= 1e-4).sum())
        flips += int((~same).sum())
    print(f"  {name}: {tot} new depths, {flips} drawn from a neighbouring bin (u on the bin edge), "
          f"{100.0 * within / tot:.2f} % within 1e-4 of the reference")
    assert within / tot >= 0.97


def test_ray_generation_vs_reference(golden_dir):
    """fneus_gen_rays_grid / fneus_gen_random_rays against Dataset.gen_rays_at / gen_random_rays_at /
    near_far_from_sphere of the reference itself (dataset.py:115-151, 186-192; tests/golden/raygen_dtu.npz), through the
    dataset class the runners use"""
    import os
    from models.dataset import SyntheticDataset
    g = dict(np.load(os.path.join(golden_dir, "raygen_dtu.npz")))
    dev = torch.device("cuda:0")
    ds = SyntheticDataset(n_images=3, H=int(g["H"]), W=int(g["W"]), device=dev)
    to = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    ds.intrinsics_all_inv, ds.pose_all = to(g["intrinsics_all_inv"]), to(g["pose_all"])      # the fixture's cameras and images
    ds.images, ds.masks = to(g["images"]), to(g["masks"])
    for lvl in (1, 4):
        o, v = ds.gen_rays_at(1, resolution_level=lvl)
        assert torch.equal(o.cpu(), torch.from_numpy(g[f"rays_at_l{lvl}/rays_o"]))
        assert (v.cpu() - torch.from_numpy(g[f"rays_at_l{lvl}/rays_v"])).abs().max().item() <= 2e-7
    for i in range(2):
        px, py = to(g[f"random_{i}/pixels_x"]), to(g[f"random_{i}/pixels_y"])
        out = ds.gen_random_rays_at(int(g[f"random_{i}/img_idx"]), len(px), pixels=(px, py))
        ref = torch.from_numpy(g[f"random_{i}/out"])
        assert torch.equal(out.cpu()[:, :3], ref[:, :3]) and torch.equal(out.cpu()[:, 6:], ref[:, 6:])     # origin, rgb, mask: copies
        assert (out.cpu()[:, 3:6] - ref[:, 3:6]).abs().max().item() <= 2e-7
        near, far = ds.near_far_from_sphere(out[:, :3], out[:, 3:6])
        assert (near.cpu() - torch.from_numpy(g[f"random_{i}/near"])).abs().max().item() <= 2e-6
        assert (far.cpu() - torch.from_numpy(g[f"random_{i}/far"])).abs().max().item() <= 2e-6
    # drawing its own pixels: in range, unit directions
    data = ds.gen_random_rays_at(0, 512)
    assert data.shape == (512, 10) and torch.allclose(data[:, 3:6].norm(dim=-1), torch.ones(512, device=dev), atol=1e-5)


@pytest.mark.parametrize("ball", [False, True])
def test_shiny_ray_generation_vs_reference(tmp_path, golden_dir, ball):
    """the Shiny-Blender feeder of config 5 end to end: files -> models/dataset.py DatasetShiny (device resident) ->
    fneus_gen_rays_grid / fneus_gen_random_rays, against the reference's own DatasetShiny on the same files
    (tests/golden/raygen_shiny.npz; dataset.py:522-662)"""
    import os
    from conftest import write_shiny_case
    from models.dataset import DatasetShiny
    g = dict(np.load(os.path.join(golden_dir, "raygen_shiny.npz")))
    tag = "ball" if ball else "disp"
    dev = torch.device("cuda:0")

    class Conf(dict):
        def get_string(self, k):
            return self[k]

    ds = DatasetShiny(Conf(data_dir=write_shiny_case(str(tmp_path / ("ball_case" if ball else "case")), g, ball)), device=dev)
    assert torch.equal(ds.masks.cpu(), torch.from_numpy(g[f"{tag}/masks"])) and torch.equal(ds.pose_all.cpu(), torch.from_numpy(g[f"{tag}/pose_all"]))
    for lvl in (1, 2):
        o, v = ds.gen_rays_at(1, resolution_level=lvl)
        assert torch.equal(o.cpu(), torch.from_numpy(g[f"{tag}/rays_at_l{lvl}/rays_o"]))
        assert (v.cpu() - torch.from_numpy(g[f"{tag}/rays_at_l{lvl}/rays_v"])).abs().max().item() <= 3e-7
    to = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    px, py = to(g[f"{tag}/random/pixels_x"]), to(g[f"{tag}/random/pixels_y"])
    out = ds.gen_random_rays_at(2, len(px), pixels=(px, py)).cpu()
    ref = torch.from_numpy(g[f"{tag}/random/out"])
    assert torch.equal(out[:, :3], ref[:, :3]) and (out[:, 3:6] - ref[:, 3:6]).abs().max().item() <= 3e-7
    assert (out[:, 6:9] - ref[:, 6:9]).abs().max().item() <= 1.2e-7 and torch.equal(out[:, 9], ref[:, 9])
    near, far = ds.near_far_from_sphere(out[:, :3].to(dev), out[:, 3:6].to(dev))
    assert (near.cpu() - torch.from_numpy(g[f"{tag}/random/near"])).abs().max().item() <= 2e-6
    assert (far.cpu() - torch.from_numpy(g[f"{tag}/random/far"])).abs().max().item() <= 2e-6


def test_embed_kernel_vs_the_torch_embedder():
    from models.embedder import Embedder, get_embedder
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for d, L in ((3, 10), (3, 4), (4, 10), (1, 4)):
        fn, width = get_embedder(L, input_dims=d)
        x = (torch.randn(1001, d, generator=g) * 2.0).to(dev)
        got = fn(x)                                  # fneus_embed
        eo = Embedder(include_input=True, input_dims=d, max_freq_log2=L - 1, num_freqs=L, log_sampling=True,
                      periodic_fns=[torch.sin, torch.cos])
        ref = eo.embed(x.cpu().double()).float()     # the torch formulation (CPU, fp64)
        assert got.shape == (1001, width) == ref.shape
        assert (got.cpu() - ref).abs().max().item() <= 2e-6 * max(1.0, 2.0 ** (L - 1) * 1e-3 + 1.0)
        xg = x.clone().requires_grad_(True)          # inputs that need a gradient stay on the torch formulation
        assert fn(xg).requires_grad


@pytest.mark.parametrize("jitter,explicit_far", [(True, False), (False, False), (True, True)])
def test_outside_depths_in_one_launch_vs_the_reference_formula(jitter, explicit_far):
    """fneus_outside_z against NeuSRenderer.render's z_vals_outside written out with torch ops exactly as the reference does
    (renderer.py:397-400, 411-419: linspace, cell mid points, jitter, flip, far / t + 1 / n_samples; far from
    dataset.py:186-192).  Tolerance: a few ulps of a value of O(1..1000) -- the kernel keeps torch's evaluation order."""
    from fneus import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    B, n_out, n_samples = 513, 32, 64
    rays_o = (torch.randn(B, 3, device=dev, generator=g) * 0.3 + torch.tensor([0.0, 0.0, -2.5], device=dev)).contiguous()
    rays_d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0], device=dev), dim=-1).contiguous()
    u = torch.rand(B, n_out, device=dev, generator=g) if jitter else None
    a = (rays_d ** 2).sum(-1, keepdim=True)
    far = 0.5 * (-2.0 * (rays_o * rays_d).sum(-1, keepdim=True)) / a + 1.0
    z = torch.linspace(1e-3, 1.0 - 1.0 / (n_out + 1.0), n_out, device=dev)
    if jitter:
        mids = 0.5 * (z[1:] + z[:-1])
        upper = torch.cat([mids, z[-1:]], -1)
        lower = torch.cat([z[:1], mids], -1)
        z = lower[None, :] + (upper - lower)[None, :] * u
    ref = far / torch.flip(z if jitter else z[None, :].expand(B, n_out), dims=[-1]) + 1.0 / n_samples
    got = ops.outside_z(rays_o, rays_d, n_out, n_samples, far=far.reshape(-1).contiguous() if explicit_far else None, u=u)
    assert got.shape == ref.shape
    rel = ((got - ref).abs() / ref.abs().clamp_min(1e-6)).max().item()
    assert rel <= 2e-6, rel
