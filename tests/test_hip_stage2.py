"""GPU parity of stage 2 (NeuSRenderer.lvis_render -> cal_indiLgt, renderer.py:567-627, calLvis.py:339-409; training step of
lvis.py:132-196) against fixtures produced by the reference itself (tests/golden/lvis_render_*.npz) and the CPU oracle.

Tolerances: 1e-4 absolute (BASELINE.json north_star) on everything that is a smooth function of the inputs.  The traced
quantities sit behind a 512-bin inverse-CDF re-sampling and a zero-crossing search, which amplify fp32 rounding at flat
stretches of the CDF (tests/test_hip_rays.py quantifies it): for them the bulk must agree to 1e-4 and the few outliers are
bounded by what the CPU oracle itself shows against the reference (tests/test_oracle_golden.py::test_stage2_lvis_render)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ["lvis_render_room_b24_n32", "lvis_render_ball_b16_n16", "lvis_render_room_b128_n64"]


def T(a):
    return torch.from_numpy(np.asarray(a))


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def build(g):
    from fneus import ops, synth
    from fneus.trainer2 import Stage2Trainer
    from fneus.trainer import WMASK_MODEL
    warp = None if int(g["warp"][0]) < 0 else (int(g["warp"][0]), float(g["warp"][1]))
    conf = dict(WMASK_MODEL, lvis_renderer=dict(n_samples=int(g["n_samples"]), n_importance=int(g["n_importance"]), n_outside=0,
                                                up_sample_steps=4, perturb=1.0))
    tr = Stage2Trainer(torch.device(DEV), model_conf=conf, prec=ops.PREC_PARITY, lr=float(g["lr"]), synthetic_init=False)
    sd = lambda d: {k: T(v).to(DEV) for k, v in d.items()}
    tr.sdf_network.load_state_dict(sd(synth.sdf_state_dict(int(g["seed_sdf"]), bias=float(g["bias"]), warp=warp,
                                                           inside_out=bool(g["room"]))))
    tr.color_network.load_state_dict(sd(synth.color_state_dict(int(g["seed_color"]))))
    tr.lvis_network.load_state_dict(sd(synth.lvis_state_dict(int(g["seed_lvis"]))))
    tr.indiLgt_network.load_state_dict(sd(synth.indilgt_state_dict(int(g["seed_indilgt"]))))
    return tr


def frac_within(a, b, tol):
    return float(((a - b).abs() <= tol).float().mean())


def test_ray_hit_and_sample_dirs_vs_oracle():
    from fneus import ops
    from oracle import ref_torch as R
    g = torch.Generator().manual_seed(3)
    B, n = 300, 96
    o = torch.randn(B, 3, generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    z = torch.sort(torch.rand(B, n, generator=g) * 1.6, dim=-1)[0]
    sdf = torch.randn(B, n, generator=g) * 0.05 + torch.linspace(0.15, -0.1, n)[None, :]
    sdf[:10] = sdf[:10].abs()                   # never negative: no hit
    sdf[10:20, 0] = -0.01                       # negative at the first sample: no hit (idx >= 1 fails)
    o[20:30] = o[20:30] + 5.0                   # far outside the unit sphere: no hit
    normal = torch.randn(B, n, 3, generator=g)
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((B, 1), 0.9 / 32)], -1)
    inv_s = 20.0
    got = ops.ray_hit(o.to(DEV), d.to(DEV), z.to(DEV), sdf.to(DEV), dists=dists.to(DEV), normal=normal.to(DEV), inv_s=inv_s,
                      want_weights=True)
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    inside = (pts.norm(dim=-1) < 1.0)
    mask, z_surf = R.first_hit(sdf, z, inside.any(-1))
    assert np.array_equal(got["sdf_mask"].cpu().numpy().astype(bool), mask.numpy())
    assert 0 < int(mask.sum()) < B
    assert (got["z_surf"].cpu()[mask] - z_surf[mask]).abs().max().item() <= 1e-5
    assert float(got["z_surf"].cpu()[~mask].abs().max()) == 0.0
    p_ref = o[mask] + d[mask] * z_surf[mask][:, None]
    assert (got["pts_surf"].cpu()[mask] - p_ref).abs().max().item() <= 1e-5
    # compute_weight (calLvis.py:119-150) on the given sdf / gradients
    tc = (d[:, None, :] * normal).sum(-1)
    ic = -torch.relu(-tc * 0.5 + 0.5)
    pc, nc = torch.sigmoid((sdf - ic * dists * 0.5) * inv_s), torch.sigmoid((sdf + ic * dists * 0.5) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)
    w = alpha * R.exclusive_transmittance(alpha)
    assert (got["weights"].cpu() - w).abs().max().item() <= 2e-6
    assert (got["occlusion"].cpu() - (w * inside.float()).sum(-1)).abs().max().item() <= 1e-5
    # the same kernel on the primary rays: explicit inside mask, no occlusion
    got2 = ops.ray_hit(o.to(DEV), d.to(DEV), z.to(DEV), sdf.to(DEV), inside_mask=inside.any(-1).to(DEV))
    assert np.array_equal(got2["sdf_mask"].cpu().numpy().astype(bool), mask.numpy())
    # sample_dirs
    M, S = 257, 4
    surf, nrm = torch.randn(M, 3, generator=g), torch.randn(M, 3, generator=g) * 3.0
    nrm[0] = torch.tensor([1.0, 0.0, 0.0])      # parallel to the frame's x axis: U degenerates to 0 / 1e-6 as in the reference
    ut, uz = torch.rand(M, S, generator=g), torch.rand(M, S, generator=g)
    origins, dirs = ops.sample_dirs(surf.to(DEV), nrm.to(DEV), ut.to(DEV), uz.to(DEV))
    ref = R.sample_dirs(nrm, ut * (2.0 * np.pi), torch.asin(uz * 0.95))
    assert (dirs.cpu().reshape(M, S, 3) - ref).abs().max().item() <= 2e-6
    assert torch.equal(origins.cpu().reshape(M, S, 3), surf[:, None, :].expand(M, S, 3))


def test_upsample_512_coarse_samples_vs_oracle():
    """the 512 -> 32 re-sampling of a secondary ray (calLvis.py:55-90): bin index and depth"""
    from fneus import ops
    from oracle import ref_torch as R
    g = torch.Generator().manual_seed(4)
    Rn, m = 200, 512
    o = torch.randn(Rn, 3, generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn(Rn, 3, generator=g), dim=-1)
    z = torch.linspace(0.0, 1.0, m)[None, :].expand(Rn, m).contiguous()
    cross = torch.rand(Rn, 1, generator=g) * 1.4              # some rays never cross (cross > 1)
    sdf = (cross - z) * 0.7 + 0.002 * torch.randn(Rn, m, generator=g)
    for inv_s in (20.085537, 512.0):
        got = ops.upsample(o.to(DEV), d.to(DEV), z.to(DEV), sdf.to(DEV), 32, inv_s).cpu()
        ref, below, cdf = R.up_sample(o, d, z, sdf, 32, inv_s, return_bins=True)
        dz = (got - ref).abs()
        # a sample lands in another bin only where the cdf is flat to 1 ulp; everywhere else the depth agrees to 1e-5
        assert (dz <= 1e-5).float().mean().item() >= 0.98, (inv_s, (dz <= 1e-5).float().mean().item())
        assert dz.max().item() <= 5e-3, (inv_s, dz.max().item())
    with pytest.raises(RuntimeError):
        ops.upsample(o.to(DEV), d.to(DEV), torch.zeros(Rn, 513, device=DEV), torch.zeros(Rn, 513, device=DEV), 32, 64.0)


@pytest.mark.parametrize("name", CASES)
def test_lvis_render_vs_reference(golden_dir, name):
    g = load(golden_dir, name)
    tr = build(g)
    data = T(g["data"]).to(DEV)
    trace = {}
    out = tr.renderer.lvis_render(data[:, :3].contiguous(), data[:, 3:6].contiguous(), T(g["near"]).to(DEV), T(g["far"]).to(DEV),
                                  u_theta=T(g["step0/u_theta"]).to(DEV), u_z=T(g["step0/u_z"]).to(DEV), trace=trace)
    c = lambda t: t.detach().cpu()
    assert np.array_equal(c(out["sdf_mask"]).numpy(), g["out/sdf_mask"])
    assert np.array_equal(c(trace["sec_sdf_mask"]).numpy(), g["trace/sec_sdf_mask"])
    m = T(g["out/sdf_mask"])
    assert (c(trace["normal"]) - T(g["trace/normal"])).abs().max().item() <= 2e-4
    assert (c(trace["dirs"]) - T(g["trace/dirs"])).abs().max().item() <= 2e-4
    zf = frac_within(c(trace["z_fine"]), T(g["trace/z_fine"]), 1e-4)
    print(f"  {name}: fine depths within 1e-4 of the reference: {100 * zf:.2f} %")
    assert zf >= 0.97
    # predictions of the two distilled networks: smooth functions of (surface point, direction)
    assert (c(out["pre_lvis"]) - T(g["out/pre_lvis"])).abs().max().item() <= 1e-4
    pr, pr_ref = c(out["pre_trace_radiance"]), T(g["out/pre_trace_radiance"])
    assert (pr - pr_ref).abs().max().item() <= 1e-4 * max(1.0, pr_ref.abs().max().item())
    # ground truth behind the re-sampling
    # (round 5: observed 100 % within 1e-4 on the three fixtures, worst 7.7e-5 / 1.2e-7; the bounds were 90 % and 2e-3)
    for key, tol_bulk, tol_max in (("gt_lvis", 1e-4, 2e-4), ("gt_trace_radiance", 1e-4, 2e-4)):
        a, b = c(out[key]), T(g["out/" + key])
        f = frac_within(a, b, tol_bulk)
        print(f"  {name}: {key} within {tol_bulk:g}: {100 * f:.2f} %, worst {float((a - b).abs().max()):.2e}")
        assert f >= 0.995 and (a - b).abs().max().item() <= tol_max, key
    assert torch.equal(c(out["gt_lvis"])[~m], torch.ones_like(c(out["gt_lvis"])[~m]))
    # loss and gradients of lvis.py:164-170 -- with the reference's own primary depths fed in (trace/prim_z): the trained networks
    # are ReLU MLPs on PE10 of the primary hit point (frequencies up to 2^9), and a zero crossing that moves by 1e-4 where the own
    # sampler's depths differ (flat-CDF stretches, tests/test_hip_rays.py) turns the first layers' high-octave columns by 5 % and
    # flips ReLUs: conditioning of the algorithm, separated from kernel error exactly as stage 1 does it
    from fneus.trainer2 import stage2_loss
    out = tr.renderer.lvis_render(data[:, :3].contiguous(), data[:, 3:6].contiguous(), T(g["near"]).to(DEV), T(g["far"]).to(DEV),
                                  u_theta=T(g["step0/u_theta"]).to(DEV), u_z=T(g["step0/u_z"]).to(DEV),
                                  z_vals_override=T(g["trace/prim_z"]).to(DEV))
    assert np.array_equal(c(out["sdf_mask"]).numpy(), g["out/sdf_mask"])
    L = stage2_loss(out)
    for k in ("loss", "lvis_loss", "trace_radiance_loss"):
        assert abs(float(L[k]) - float(g["step0/" + k])) <= 1e-3 * max(1.0, abs(float(g["step0/" + k]))), k
    L["loss"].backward()
    named = [("lvis." + k, p) for k, p in tr.lvis_network.named_parameters()] + \
            [("indi." + k, p) for k, p in tr.indiLgt_network.named_parameters()]
    for k, prm in named:
        ref_sub, ref_norm = g["grad_sub/" + k], float(g["grad_norm/" + k])
        sub = prm.grad.detach().cpu().reshape(-1)[::997].numpy()
        scale = max(ref_norm / np.sqrt(prm.numel()), np.abs(ref_sub).max(), 1e-7)
        assert np.abs(sub - ref_sub).max() <= 2e-2 * scale + 1e-7, (k, np.abs(sub - ref_sub).max(), scale)
        assert abs(prm.grad.double().norm().item() - ref_norm) <= 1e-2 * ref_norm + 1e-7, k


@pytest.mark.parametrize("name", CASES)
def test_secondary_march_teacher_forced(golden_dir, name):
    """The traced ground truth with the reference's OWN fine depths fed in (trace/z_fine; the secondary rays' origins and
    directions from the fixture as well): everything behind the 512 -> 32 inverse-CDF re-sampling -- SDF + gradient at the 32
    mid points (K2), NeuS weights / occlusion and first hit (fneus_ray_hit), colour at the hit point (K2 + K4) -- must meet the
    1e-4 of BASELINE.json on EVERY secondary ray, as stage 1 does with teacher-forced depths.  (With its own re-sampling the
    path differs at a few flat-CDF outliers: test_lvis_render_vs_reference.)"""
    from models.calLvis import _secondary_march, frozen_inv_s
    g = load(golden_dir, name)
    tr = build(g)
    origins = T(g["trace/sec_origins"]).to(DEV).contiguous()
    dirs = T(g["trace/dirs"]).reshape(-1, 3).to(DEV).contiguous()
    assert origins.shape == dirs.shape
    trace = {}
    with torch.no_grad():
        inv_s = frozen_inv_s(tr.deviation_network)
        assert abs(inv_s - float(g["trace/inv_s"])) <= 1e-4 * inv_s
        occu, hit_rgb, mask = _secondary_march(origins, dirs, tr.sdf_network, tr.color_network, inv_s, trace,
                                               z_fine_override=T(g["trace/z_fine"]).to(DEV))
    c = lambda t: t.detach().cpu()
    assert np.array_equal(c(mask).numpy().astype(bool), g["trace/sec_sdf_mask"])
    e_w = (c(trace["sec_weights"]) - T(g["trace/sec_weights"])).abs().max().item()
    e_rgb = (c(hit_rgb) - T(g["trace/sec_hit_rgb"])).abs().max().item()
    m = T(g["out/sdf_mask"])
    gt_lvis = T(g["out/gt_lvis"])[m].reshape(-1)
    e_vis = ((1.0 - c(occu)) - gt_lvis).abs().max().item()
    print(f"  {name}: teacher-forced secondary march: weights {e_w:.1e}, visibility {e_vis:.1e}, hit colour {e_rgb:.1e}")
    assert e_w <= 1e-4 and e_vis <= 1e-4 and e_rgb <= 1e-4


def test_stage2_adam_steps_match_reference(golden_dir):
    """three training steps of lvis.py:132-196 on the reference's own direction draws"""
    g = load(golden_dir, "lvis_render_room_b24_n32")
    tr = build(g)
    data = T(g["data"]).to(DEV)
    named = [("lvis." + k, p) for k, p in tr.lvis_network.named_parameters()] + \
            [("indi." + k, p) for k, p in tr.indiLgt_network.named_parameters()]
    lr = float(g["lr"])
    for step in range(3):
        L = tr.train_step(data, near=T(g["near"]).to(DEV), far=T(g["far"]).to(DEV),
                          u_theta=T(g[f"step{step}/u_theta"]).to(DEV), u_z=T(g[f"step{step}/u_z"]).to(DEV),
                          z_vals_override=T(g["trace/prim_z"]).to(DEV))      # the reference's primary depths (see above)
        ref = float(g[f"step{step}/loss"])
        assert abs(float(L["loss"]) - ref) <= 2e-3 * max(1.0, abs(ref)), (step, float(L["loss"]), ref)
        if step in (0, 2):
            for k, prm in named:
                want = g[f"adam{step + 1}_sub/" + k]
                got = prm.detach().cpu().reshape(-1)[::997].numpy()
                bad = np.abs(got - want) > 0.2 * lr       # Adam moves every weight by ~lr: near-zero gradients may flip sign
                # ... and the traced ground truth differs at a few re-sampling outliers (see above): after three steps a few
                # weights in a hundred have drifted by more than 0.2 lr, none by more than two flipped signs
                print(f"  adam step {step + 1} {k}: {int(bad.sum())} of {bad.size} sampled weights off by > 0.2 lr, worst {np.abs(got - want).max() / lr:.2f} lr")
                # (round 5: observed none after one step -- worst 0.04 lr -- and <= 3.0 % after three (2 of the 33 sampled entries of the
                # smallest tensor), worst 2.67 lr; the bounds
                # were 2 % / 10 % and 2.2 lr per step)
                assert bad.sum() <= max(1 if step == 0 else 2, (0.005 if step == 0 else 0.05) * bad.size), (step, k, int(bad.sum()), bad.size)
                assert np.abs(got - want).max() <= (0.2 if step == 0 else 4.0) * lr + 1e-7, (step, k)   # a flipped sign is 2 lr per step
    assert tr.iter_step == 3


def test_stage2_step_at_full_size_properties():
    """config 3 shape: 512 rays x (64 + 64), <= 2048 secondary rays x 512 coarse samples.  Size-independent properties:
    visibility in [0, 1], rows without a hit are exactly 1, the step is finite and lowers the loss on a fixed batch."""
    from fneus.trainer import synthetic_batches
    from fneus.trainer2 import Stage2Trainer
    tr = Stage2Trainer(torch.device(DEV), seed=3)
    batch = synthetic_batches(1, 512, torch.device(DEV), seed0=11)[0]
    g = torch.Generator(device=DEV).manual_seed(0)
    first = last = None
    for i in range(12):
        out = tr.train_step(batch)
        assert out is not None and bool(torch.isfinite(out["loss"]))
        first = float(out["loss"]) if first is None else first
        last = float(out["loss"])
    assert int(out["n_hit"]) > 100
    assert last < first, (first, last)
    o, d = batch[:, :3].contiguous(), batch[:, 3:6].contiguous()
    res = tr.renderer.lvis_render(o, d, None, None)
    gt = res["gt_lvis"]
    assert float(gt.min()) >= -1e-5 and float(gt.max()) <= 1.0 + 1e-5
    miss = ~res["sdf_mask"]
    for k in ("gt_lvis", "pre_lvis", "gt_trace_radiance", "pre_trace_radiance"):
        assert bool((res[k][miss] == 1.0).all()), k


def test_fixed_shape_render_matches_the_reference_and_the_graph_step_trains(golden_dir):
    """lvis_render(fixed_shape=True) -- every ray treated as a hit point, masked afterwards -- gives the reference's outputs
    when the reference's draws are put on the rows of its hit points; the trainer's hipGraph mode replays that step"""
    from fneus.trainer import synthetic_batches
    from fneus.trainer2 import Stage2Trainer, stage2_loss
    g = load(golden_dir, "lvis_render_room_b24_n32")
    tr = build(g)
    data = T(g["data"]).to(DEV)
    m = T(g["out/sdf_mask"])
    B = len(m)
    ut, uz = torch.full((B, 4), 0.5), torch.full((B, 4), 0.5)
    ut[m], uz[m] = T(g["step0/u_theta"]), T(g["step0/u_z"])
    out = tr.renderer.lvis_render(data[:, :3].contiguous(), data[:, 3:6].contiguous(), T(g["near"]).to(DEV), T(g["far"]).to(DEV),
                                  u_theta=ut.to(DEV), u_z=uz.to(DEV), fixed_shape=True)
    c = lambda t: t.detach().cpu()
    assert np.array_equal(c(out["sdf_mask"]).numpy(), g["out/sdf_mask"])
    assert (c(out["pre_lvis"]) - T(g["out/pre_lvis"])).abs().max().item() <= 1e-4
    assert (c(out["pre_trace_radiance"]) - T(g["out/pre_trace_radiance"])).abs().max().item() <= 1e-4 * max(1.0, float(np.abs(g["out/pre_trace_radiance"]).max()))
    for key in ("gt_lvis", "gt_trace_radiance"):
        a, b = c(out[key]), T(g["out/" + key])
        assert frac_within(a, b, 1e-4) >= 0.9 and (a - b).abs().max().item() <= 2e-3, key
    L = stage2_loss(out)
    assert abs(float(L["loss"].detach()) - float(g["step0/loss"])) <= 1e-3 * max(1.0, abs(float(g["step0/loss"])))
    trg = Stage2Trainer(torch.device(DEV), seed=3, use_graph=True)
    batch = synthetic_batches(1, 512, torch.device(DEV), seed0=11)[0]
    losses = [float(trg.train_step(batch)["loss"]) for _ in range(12)]
    assert trg._graph is not None and trg.iter_step == 12
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    from fneus import synth
    away = torch.from_numpy(synth.ray_batch(512, seed=5, n_miss=512)).to(DEV)        # every ray misses the unit sphere
    o = trg.train_step(away)
    assert int(o["n_hit"]) == 0 and float(o["loss"]) == 0.0
    assert all(bool(torch.isfinite(p).all()) for p in trg.params)


def test_frozen_weights_loaded_after_the_capture_reach_the_replayed_step():
    """The captured step holds no pack launch for a frozen network (it was packed by the eager warm-up); a load_state_dict into the
    frozen SDF network AFTER the capture must still reach the replays (Stage2Trainer._refresh_frozen packs in front of the replay)."""
    from fneus import synth
    from fneus.trainer import synthetic_batches
    from fneus.trainer2 import Stage2Trainer
    dev = torch.device(DEV)
    batch = synthetic_batches(1, 256, dev, seed0=21)[0]
    trg = Stage2Trainer(dev, seed=3, use_graph=True)
    for _ in range(4):
        trg.train_step(batch)
    assert trg._graph is not None
    hits_before = int(trg.train_step(batch)["n_hit"])
    new_sd = {k: torch.from_numpy(v) for k, v in synth.sdf_state_dict(9).items()}
    trg.sdf_network.load_state_dict(new_sd)                     # a different surface
    o = trg.train_step(batch)
    tre = Stage2Trainer(dev, seed=3, use_graph=False)
    tre.sdf_network.load_state_dict(new_sd)
    rays_o, rays_d = batch[:, :3].contiguous(), batch[:, 3:6].contiguous()
    ref = tre.renderer.lvis_render(rays_o, rays_d, None, None, fixed_shape=True)
    n_new = int(ref["sdf_mask"].sum())
    assert n_new != hits_before, "the two synthetic surfaces must differ in their hit count for this test to have teeth"
    assert int(o["n_hit"]) == n_new                             # the replay saw the new surface


def test_fused_indirect_radiance_vs_the_element_wise_formulation():
    """IndirectLight.radiance (fneus_indir_illum_fwd / _bwd: the network's output transform + query_indir_illum in one launch each)
    against query_indir_illum(IndirectLight.forward(.)) through autograd: values and the gradients of every parameter"""
    from fneus import synth
    from models.calLvis import query_indir_illum
    from models.fields import IndirectLight
    dev = torch.device(DEV)
    net = IndirectLight()
    net.load_state_dict({k: T(v) for k, v in synth.indilgt_state_dict(5).items()})
    net.to(dev)
    g = torch.Generator().manual_seed(2)
    n, S = 301, 4
    pts = (torch.randn(n, 3, generator=g) * 0.4).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, S, 3, generator=g), dim=-1).to(dev)
    cot = torch.randn(n, S, 3, generator=g).to(dev)

    def run(fused):
        for p in net.parameters():
            p.grad = None
        rad = net.radiance(pts, dirs) if fused else query_indir_illum(net(pts), dirs)
        (rad * cot).sum().backward()
        return rad.detach().clone(), [p.grad.detach().clone() for p in net.parameters()]

    r0, g0 = run(False)
    r1, g1 = run(True)
    assert (r1 - r0).abs().max().item() <= 2e-6 * max(1.0, r0.abs().max().item())
    for a, b in zip(g1, g0):
        assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6)
