"""GPU parity of the whole NeuSRenderer.render (HIP backend) against fixtures produced by the reference itself
(tests/golden/render_*.npz) and against the CPU oracle.  Tolerances: 1e-4 absolute on colours / weights / sdf in
parity mode (BASELINE.json north_star); gradients relative to each tensor's scale."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RAY_KEYS = ("color_fine", "surface_color", "weight_sum", "gradient_error", "specular_color", "diffuse_color", "s_val")
SAMPLE_KEYS = ("cdf_fine", "weight_max", "gradients", "weights", "inside_sphere")


def T(a):
    return torch.from_numpy(np.asarray(a))


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def stored(x, g):
    """restrict a per-sample array ([B, n, ...] or [B*n, ...]) of a full-batch run to the rays a strided fixture stores"""
    s = int(g["ray_stride"]) if "ray_stride" in g else 1
    if s == 1:
        return x
    B = int(g["B"])
    x = x.detach()
    if x.shape[0] == B:
        return x[::s]
    return x.reshape((B, -1) + tuple(x.shape[1:]))[::s].reshape((-1,) + tuple(x.shape[1:]))


def final_z(g):
    return T(g["trace/z_final"] if "trace/z_final" in g else g["trace/z_3"])


def build(g, prec, gprec=None):
    from fneus import synth
    from models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork, RefColor, NeRF
    from models.renderer import NeuSRenderer
    sdf = SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                     geometric_init=True, weight_norm=True)
    col = RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=4, weight_norm=True,
                           multires_view=4, squeeze_out=True)
    var = SingleVarianceNetwork(0.3)
    ref = RefColor()
    sdf.load_state_dict({k: T(v) for k, v in synth.sdf_state_dict(int(g["seed_sdf"])).items()})
    col.load_state_dict({k: T(v) for k, v in synth.color_state_dict(int(g["seed_color"])).items()})
    ref.load_state_dict({k: T(v) for k, v in synth.refcolor_state_dict(int(g["seed_refcolor"])).items()})
    for m in (sdf, col, var, ref):
        m.to(DEV)
    sdf.set_precision(prec)
    col.set_precision(prec)
    for m in (sdf, col, ref):
        m.set_gradient_precision(gprec)
    nerf = None
    if int(g["n_outside"]) > 0:
        nerf = NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True)
        nerf.load_state_dict({k: T(v) for k, v in synth.nerf_state_dict(int(g["seed_nerf"])).items()})
        nerf.to(DEV)
        nerf.set_gradient_precision(gprec)
    rnd = NeuSRenderer(int(g["n_samples"]), int(g["n_importance"]), int(g["n_outside"]), 4, 1.0, nerf=nerf,
                       sdf_network=sdf, deviation_network=var, color_network=col, refColor_network=ref)
    return rnd, dict(sdf=sdf, color=col, var=var, refcolor=ref, nerf=nerf)


def run(g, prec, teacher_z, fused_loss=False, gprec=None):
    from oracle import ref_torch as R
    rnd, nets = build(g, prec, gprec)
    data = T(g["data"]).to(DEV)
    rays_o, rays_d, rgb, mask = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
    near, far = R.near_far_from_sphere(rays_o, rays_d)
    bg = torch.ones(1, 3, device=DEV) if int(g["white_bkgd"]) else None
    out = rnd.render(rays_o, rays_d, near, far, perturb_overwrite=0, cos_anneal_ratio=float(g["cos_anneal_ratio"]),
                     background_rgb=bg, z_vals_override=final_z(g).to(DEV) if teacher_z else None,
                     loss_args=(rgb, mask, 0.1, float(g["mask_weight"]), 0.1) if fused_loss else None)
    return out, nets, (rgb, mask)


def maxerr(a, b):
    return (a.detach().cpu().double().reshape(-1) - T(b).double().reshape(-1)).abs().max().item()


WMASK = ["render_wmask_b16_n16", "render_wmask_b8_n64", "render_wmask_b16_n16_c0"]
WOMASK = ["render_womask_b16_n16_o8",      # n_outside = 8, white background, cos_anneal 0.3
          "render_womask_b64_n64_o32"]     # womask.conf's sample counts (64 + 64 + 32), no background colour, cos_anneal 0.7
# round 2: BASELINE config 1 (256 rays x (32+32), 8 new depths per step) and 64 rays at the reference depth (64+64)
# ... and BASELINE config 2 at full size (512 rays x (64+64): the shape bench.py measures)
BIG = ["render_wmask_b256_n32", "render_wmask_b64_n64", "render_wmask_b512_n64"]


@pytest.mark.parametrize("name", WMASK + WOMASK + BIG)
def test_render_core_teacher_forced(golden_dir, name):
    g = load(golden_dir, name)
    out, _, _ = run(g, 3, teacher_z=True)
    assert np.array_equal(out["sdf_mask"].cpu().numpy(), g["out/sdf_mask"])
    for k in RAY_KEYS + SAMPLE_KEYS:
        x = out[k]
        x = stored(x, g) if (x.dim() >= 2 and x.shape[1] not in (1, 3)) else x      # per-sample arrays: stored rays only
        e = maxerr(x, g["out/" + k])
        assert e <= 1e-4, (k, e)
    assert maxerr(stored(out["_sdf"], g), g["core/sdf"]) <= 1e-4
    assert maxerr(stored(out["_mid_z_vals"], g), g["core/mid_z_vals"]) <= 1e-6


@pytest.mark.parametrize("name", WMASK + BIG)
def test_render_end_to_end(golden_dir, name):
    """own sampler: ray-integrated outputs within 1e-4 (per-sample outputs sit at slightly different z: the inverse
    CDF is ill-conditioned where the pdf is flat, see tests/test_oracle_golden.py)"""
    g = load(golden_dir, name)
    out, _, _ = run(g, 3, teacher_z=False)
    assert np.array_equal(out["sdf_mask"].cpu().numpy(), g["out/sdf_mask"])
    # 64+64 samples (the reference configuration): 1e-4.  The 16+16 toy cases integrate so coarsely that a z shift
    # of 1e-4 moves a ray integral by a few 1e-4 -- the reference shows the same sensitivity against its own
    # restatement (tests/test_oracle_golden.py::test_render_end_to_end uses the same split).
    tol = 1e-4 if int(g["n_samples"]) >= 64 else 5e-4
    errs = {k: maxerr(out[k], g["out/" + k]) for k in RAY_KEYS}
    print(name, {k: f"{v:.1e}" for k, v in errs.items()}, "z", f"{maxerr(out['_z_vals'], final_z(g)):.1e}")
    # Rays beyond the tolerance must be rare, must be rays with a depth that moved by more than 1e-3 (drawn from a flat cdf
    # bin, where the inverse cdf amplifies fp32 rounding -- tests/test_hip_rays.py separates that from bin choices -- so the
    # ray is integrated over other points) and must stay within 5e-4: the reference's own restatement on the CPU differs from
    # the reference by 1.4e-4 in weight_sum on one such ray of the 512-ray fixture.
    moved = ((out["_z_vals"].detach().cpu() - final_z(g)).abs() > 1e-3).any(dim=1)
    for k in RAY_KEYS:
        if out[k].dim() >= 1 and out[k].shape[0] == moved.shape[0]:
            d = (out[k].detach().cpu().double() - T(g["out/" + k]).double()).abs().reshape(moved.shape[0], -1).max(dim=1).values
            bad = d > tol
            assert int(bad.sum()) <= max(1, moved.shape[0] // 256) and bool(moved[bad].all()) and d.max().item() <= 5e-4, \
                (k, int(bad.sum()), d.max().item())
        else:
            assert errs[k] <= tol, (k, errs[k])
    # depths: the inverse cdf is ill conditioned where the pdf is flat (tests/test_hip_rays.py separates that from bin
    # choices on the reference's own sampler trace): a handful of depths in flat bins move by 1e-3..1e-2
    dz = (out["_z_vals"].detach().cpu() - final_z(g)).abs()
    frac = (dz <= 1e-4).float().mean().item()
    print(name, f"depths within 1e-4: {100 * frac:.2f} %, worst {dz.max().item():.1e}")
    # (sorted arrays: one moved depth shifts its neighbours' slots.)  At the reference's sample counts (64 + 64) >= 97 % of the
    # depths of a batch agree to 1e-4 (observed: 97.66 % of the 512-ray fixture, 99.06 % of the 64-ray one); a single ray with a moved
    # depth is 1/8 of the 8-ray fixture (94.6 %), and the toy depths integrate a coarser cdf (97.9 ... 99.6 %)
    big = int(g["n_samples"]) >= 64 and int(g["B"]) >= 64
    assert dz.max().item() <= (3e-3 if int(g["B"]) <= 16 else 2e-2) and frac >= (0.97 if big else 0.9)


@pytest.mark.parametrize("gprec", [3, 2, 1], ids=["grad_hi_lo", "grad_mixed", "grad_bf16"])
@pytest.mark.parametrize("fused", [False, True], ids=["torch_loss", "fused_loss"])
@pytest.mark.parametrize("name", WMASK[:2] + WOMASK + BIG)
def test_loss_and_gradients(golden_dir, name, fused, gprec):
    """fused: shading + blend + losses + their gradients from fneus_stage1_loss (what the training step uses);
    otherwise the same terms written with torch ops on the render dict.
    gprec 3: the backward stash holds hi + lo planes (fp32-accurate weight gradients); gprec 1 (the training default):
    bf16 planes -- every product of the weight-gradient GEMMs carries 2^-9 rounding, which shows where a sum cancels
    (bias gradients of 3 outputs over 512 samples); gprec 2: bf16 planes but for the colour network's output layer (the tensors
    that showed it: tools/experiments/r05/gprec_tensors.py) -- held to the bounds of gprec 3."""
    from _helper_losses import stage1_loss
    g = load(golden_dir, name)
    out, nets, (rgb, mask) = run(g, 3, teacher_z=True, fused_loss=fused, gprec=gprec)
    if fused:
        losses = out["losses"]
        assert maxerr(out["surface_color"], g["out/surface_color"]) <= 1e-4
    else:
        losses = stage1_loss(out, rgb, mask, igr_weight=0.1, mask_weight=float(g["mask_weight"]), surface_weight=0.1)
    for k, key in (("loss", "loss"), ("color", "color_loss"), ("surface", "surface_loss"), ("eikonal", "eikonal_loss"),
                   ("mask", "mask_loss")):
        assert abs(losses[key].item() - float(g["loss/" + k])) <= 1e-4, k
    losses["loss"].backward()
    checked = 0
    worst = 0.0
    for key in g:
        if not key.startswith("grad_norm/"):
            continue
        pname = key[len("grad_norm/"):]
        net, rest = pname.split(".", 1)
        if nets.get(net) is None:
            continue
        prm = dict(nets[net].named_parameters())[rest]
        assert prm.grad is not None, pname
        ref_norm = float(g[key])
        ref_sub = g["grad_sub/" + pname]
        sub = prm.grad.detach().cpu().reshape(-1)[::997].numpy()
        scale = max(ref_norm / np.sqrt(prm.numel()), np.abs(ref_sub).max(), 1e-7)
        e_sub = np.abs(sub - ref_sub).max() / scale
        e_norm = abs(prm.grad.double().norm().item() - ref_norm) / (ref_norm + 1e-12)
        worst = max(worst, e_sub, e_norm)
        # RefColor is a ReLU MLP evaluated on only 2 samples per masked ray: a single unit whose pre-activation sits
        # within rounding of zero shows up in an individual weight-gradient entry
        lim_sub = 3e-2 if net in ("refcolor", "nerf") else (5e-3 if gprec in (2, 3) else 8e-3)
        # background NeRF (K7): 9 ReLU layers on only 16 x 24 samples -- one unit at a ReLU boundary is 0.5 % of a norm
        # (tests/test_hip_nerf.py masks such samples and sees 1e-5)
        lim_norm = 1e-2 if net == "nerf" else (2e-3 if gprec in (2, 3) else 5e-3)
        assert e_sub <= lim_sub and e_norm <= lim_norm, (pname, e_sub, e_norm)
        checked += 1
    print(f"{name} gprec={gprec}: {checked} parameter tensors, worst relative gradient error {worst:.2e}")
    assert checked >= 40


def test_fast_mode_reports_error(golden_dir):
    """bf16 fast mode: not a parity mode; its observed error is printed and loosely bounded"""
    g = load(golden_dir, "render_wmask_b8_n64")
    out, _, _ = run(g, 1, teacher_z=True)
    errs = {k: maxerr(out[k], g["out/" + k]) for k in ("color_fine", "weights", "gradients", "weight_sum")}
    print("bf16 fast mode max abs errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["color_fine"] <= 5e-2 and errs["weight_sum"] <= 5e-2


# gradient precision 1 (bf16 planes for the weight-gradient GEMM) is what bench.py times; 3 (hi + lo planes) is fp32-accurate.
# Bounds per mode: (fraction of elements off by > lr/4 after 1 / 3 steps, median |difference| in units of lr)
# Observed (MI355X, four fixtures, the same on two runs; tools/runs/r04_t.sh): after 1 step no element off in either mode, median
# <= 1e-4 lr; after 3 steps gprec 3: <= 0.0303 of a tensor's sampled elements (one of 33), median <= 0.0174 lr; gprec 1: <= 0.0400,
# median <= 0.0177 lr.  Bounds = observed + 50 %.
ADAM_BOUNDS = {3: (0.01, 0.046, 0.026), 1: (0.01, 0.06, 0.027)}


@pytest.mark.parametrize("gprec", [3, 1])
@pytest.mark.parametrize("name", ["render_wmask_b16_n16", "render_wmask_b64_n64", "render_wmask_b256_n32", "render_wmask_b512_n64"])
def test_adam_steps_match_reference(golden_dir, name, gprec):
    """Parameters after 1 and 3 optimiser steps of the reference loop (exp_runner.py:179-181: zero_grad, backward, Adam.step
    on the fixture batch) -- adam1_sub / adam3_sub of the fixtures -- against the training step of this repo: own sampler,
    fused loss, weight gradients, fneus_adam.  Adam's first update is lr * g / (|g| + eps): +-lr wherever the gradient is
    not noise, so the comparison is per element against lr, not against the parameter's size."""
    from fneus import ops
    from fneus.trainer import Stage1Trainer, WMASK_MODEL
    import copy
    g = load(golden_dir, name)
    conf = copy.deepcopy(WMASK_MODEL)
    conf["neus_renderer"] = dict(n_samples=int(g["n_samples"]), n_importance=int(g["n_importance"]), n_outside=0,
                                 up_sample_steps=4, perturb=0.0)
    lr = 5e-4
    tr = Stage1Trainer(torch.device(DEV), model_conf=conf, prec=ops.PREC_PARITY, seed=int(g["seed_sdf"]), lr=lr,
                       mask_weight=float(g["mask_weight"]), use_graph=False, gprec=gprec)
    b1, b3, bmed = ADAM_BOUNDS[gprec]
    assert int(g["seed_color"]) == int(g["seed_sdf"]) + 1 and int(g["seed_refcolor"]) == int(g["seed_sdf"]) + 2
    data = T(g["data"]).to(DEV)
    nets = dict(sdf=tr.sdf_network, color=tr.color_network, var=tr.deviation_network)
    p0 = {f"{n}.{k}": v.detach().clone() for n, m in nets.items() for k, v in m.named_parameters()}
    for step in (1, 2, 3):
        losses = tr.train_step(data, cos_anneal_ratio=float(g["cos_anneal_ratio"]))
        if step == 1:
            assert abs(losses["loss"].item() - float(g["adam1_loss"])) <= 2e-4
        if step not in (1, 3):
            continue
        worst_frac, worst_med, checked = 0.0, 0.0, 0
        for key in g:
            if not key.startswith(f"adam{step}_sub/"):
                continue
            pname = key.split("/", 1)[1]
            net, rest = pname.split(".", 1)
            prm = dict(nets[net].named_parameters())[rest]
            got = prm.detach().cpu().reshape(-1)[::997].numpy()
            ref, start = g[key], p0[pname].cpu().reshape(-1)[::997].numpy()
            moved = np.abs(ref - start)
            if step == 1:
                assert moved.max() <= lr * 1.001 + 1e-7, pname              # the reference itself moved by <= lr
            off = np.abs(got - ref) > 0.25 * lr                             # an element whose update went elsewhere
            worst_frac = max(worst_frac, off.mean())
            # elements off by more than lr/4: gradient at rounding level (either sign is "right"); a few per thousand
            assert off.mean() <= (b1 if step == 1 else b3), (pname, step, off.mean())
            if got.size >= 20:
                worst_med = max(worst_med, float(np.median(np.abs(got - ref))) / lr)
                assert np.median(np.abs(got - ref)) <= bmed * lr, (pname, step)
            checked += 1
        assert checked >= 40
        print(f"  {name} gprec {gprec}: after {step} Adam step(s) {checked} tensors; worst fraction of elements off by > lr/4: {worst_frac:.4f}, "
              f"worst median |difference| {worst_med:.4f} lr")


def test_lvis_render_util_vs_reference(golden_dir):
    """NeuSRenderer.lvis_mateIllu_render_util (renderer.py:503-564, the entry of the stage-2 / stage-3 renderers) against the
    reference's own output: K1 + K6 only"""
    from oracle import ref_torch as R
    g = load(golden_dir, "lvis_util_b24_n32")
    g.setdefault("n_outside", 0)
    rnd, _ = build(g, 3)
    data = T(g["data"]).to(DEV)
    near, far = R.near_far_from_sphere(data[:, :3], data[:, 3:6])
    out = rnd.lvis_mateIllu_render_util(data[:, :3].contiguous(), data[:, 3:6].contiguous(), near, far)
    assert int(out["n_samples"]) == int(g["out/n_samples"])
    assert np.array_equal(out["inside_sphere_mask"].cpu().numpy(), g["out/inside_sphere_mask"])
    dz = (out["mid_z_vals"].cpu() - T(g["out/mid_z_vals"])).abs()
    ds = (out["sdf"].cpu().reshape(-1) - T(g["out/sdf"]).reshape(-1)).abs()
    print(f"  lvis util: mid_z max {dz.max():.1e} median {dz.median():.1e}; sdf max {ds.max():.1e} median {ds.median():.1e}")
    # own sampler: a few depths in flat-pdf bins move (see test_render_end_to_end); the sdf at the same depth is within 1e-4
    assert dz.max().item() <= 1e-2 and (dz <= 1e-4).float().mean().item() >= 0.9
    same = (dz <= 1e-6).reshape(-1)
    assert ds[same].max().item() <= 1e-4


def test_outside_select_lists_the_samples_render_core_uses():
    """fneus_outside_select: in the list <=> the sample is a background sample (i >= n) or its section mid point is not inside
    the unit sphere (the `inside_sphere` render_core multiplies the foreground with, renderer.py:270-272, 350-356); the rows of
    the list are the rows fneus_outside_points makes for those samples, in ray-major order."""
    from fneus import ops, synth
    B, n, n_out = 96, 128, 32
    data = torch.from_numpy(synth.ray_batch(B, seed=5, half_extent=0.9)).to(DEV)        # rays up to the sphere's rim
    o, d = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    mid = -(o * d).sum(-1, keepdim=True) / (d * d).sum(-1, keepdim=True)
    g = torch.Generator(device="cpu").manual_seed(3)
    z_core = (mid - 1.0 + 2.0 * torch.rand(B, n, generator=g).sort(dim=1).values.to(DEV)).contiguous()
    z_out = (mid + 1.0 + 0.1 + 3.0 * torch.rand(B, n_out, generator=g).sort(dim=1).values.to(DEV)).contiguous()
    z_feed = torch.cat([z_core, z_out], dim=1).contiguous()
    sd = 2.0 / 64
    s = ops.outside_select(o, d, z_core, z_feed, sd)
    cnt = int(s.count.item())
    sel = s.sel[:cnt].long().cpu()
    assert bool((sel[1:] > sel[:-1]).all())                                                # ray-major, each sample once
    listed = torch.zeros(B * (n + n_out), dtype=torch.bool)
    listed[sel] = True
    listed = listed.reshape(B, n + n_out)
    assert bool(listed[:, n:].all())
    _, mid_z = ops.sections(z_core, sd)
    pn = (o[:, None, :] + d[:, None, :] * mid_z[..., None]).norm(dim=-1).cpu()
    assert bool(listed[:, :n][pn >= 1.0].all())                                            # every sample render_core takes from it
    assert not bool(listed[:, :n][pn < 1.0 - 1e-4].any())                                  # and nothing from well inside
    frac = listed.float().mean().item()
    assert 0.2 < frac < 0.9, frac
    pts4, dirs, dists = ops.outside_points(o, d, z_feed, sd)
    k = sel.to(DEV)
    assert torch.equal(s.pts4[:cnt], pts4[k]) and torch.equal(s.dirs[:cnt], dirs[k]) and torch.equal(s.dists[:cnt], dists.reshape(-1)[k])
    assert bool((s.alpha_full.reshape(-1).cpu()[~listed.reshape(-1)] == 0).all())
    assert bool((s.rgb_full.reshape(-1, 3).cpu()[~listed.reshape(-1)] == 0).all())


@pytest.mark.parametrize("name", WOMASK)
def test_background_only_where_it_is_used_changes_nothing(golden_dir, name, monkeypatch):
    """womask: the background network evaluated at the listed samples only (the default) against all n + n_out depths per ray
    as the reference does it (FNEUS_BG_SELECT=0): inside the unit sphere its value is multiplied by exactly 0, so the rendered
    outputs are the same numbers and the NeRF's weight gradients the same sums in another order."""
    from _helper_losses import stage1_loss
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FNEUS_BG_SELECT", mode)
        g = load(golden_dir, name)
        out, nets, (rgb, mask) = run(g, 3, teacher_z=True, gprec=3)
        L = stage1_loss(out, rgb, mask, igr_weight=0.1, mask_weight=float(g["mask_weight"]), surface_weight=0.1)
        L["loss"].backward()
        res[mode] = (out, {k: p.grad.detach().clone() for k, p in nets["nerf"].named_parameters()},
                     {k: p.grad.detach().clone() for k, p in nets["sdf"].named_parameters()})
    a, b = res["1"], res["0"]
    for k in ("color_fine", "weights", "weight_sum"):
        assert torch.equal(a[0][k], b[0][k]), k
    for grads in (1, 2):
        for k in a[grads]:
            scale = b[grads][k].abs().max().item() + 1e-12
            assert (a[grads][k] - b[grads][k]).abs().max().item() <= 2e-5 * scale + 1e-9, (k, scale)


@pytest.mark.parametrize("name", ["render_wmask_b64_n64"])
def test_colour_products_inside_the_sdf_launch_change_nothing(golden_dir, name, monkeypatch):
    """the colour network's weight-gradient products in the SDF network's launch behind K3 (default) against a launch of their own
    behind the colour backward (FNEUS_GEMM_MERGE=0): the same sums over the same planes, split over workgroups differently"""
    from _helper_losses import stage1_loss
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FNEUS_GEMM_MERGE", mode)
        g = load(golden_dir, name)
        out, nets, (rgb, mask) = run(g, 3, teacher_z=True, gprec=1)
        L = stage1_loss(out, rgb, mask, igr_weight=0.1, mask_weight=float(g["mask_weight"]), surface_weight=0.1)
        L["loss"].backward()
        res[mode] = {f"{n}.{k}": p.grad.detach().clone() for n in ("sdf", "color") for k, p in nets[n].named_parameters()}
    for k in res["1"]:
        scale = res["0"][k].abs().max().item() + 1e-12
        assert (res["1"][k] - res["0"][k]).abs().max().item() <= 2e-5 * scale + 1e-9, (k, scale)


@pytest.mark.parametrize("name", ["render_wmask_b64_n64", "render_womask_b16_n16_o8"])
def test_two_renders_one_backward_keep_every_weight_gradient(golden_dir, name, monkeypatch):
    """two differentiable renders of different batch size before ONE backward pass (what the stash-overwrite message recommends):
    the first render's colour / RefColor / background products wait for a weight-gradient launch that is stamped for the FIRST
    SDF forward, not for the latest one -- merged launches (default) against launches of their own (FNEUS_GEMM_MERGE=0)"""
    from oracle import ref_torch as R
    from _helper_losses import stage1_loss
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FNEUS_GEMM_MERGE", mode)
        g = load(golden_dir, name)
        rnd, nets = build(g, 3, 1)
        data = T(g["data"]).to(DEV)
        bg = torch.ones(1, 3, device=DEV) if int(g["white_bkgd"]) else None
        total = 0.0
        for rows in (data[: data.shape[0] // 2], data):
            rays_o, rays_d, rgb, mask = rows[:, :3], rows[:, 3:6], rows[:, 6:9], rows[:, 9:10]
            near, far = R.near_far_from_sphere(rays_o, rays_d)
            out = rnd.render(rays_o, rays_d, near, far, perturb_overwrite=0, cos_anneal_ratio=float(g["cos_anneal_ratio"]),
                             background_rgb=bg)
            total = total + stage1_loss(out, rgb, mask, igr_weight=0.1, mask_weight=float(g["mask_weight"]),
                                        surface_weight=0.1)["loss"]
        total.backward()
        res[mode] = {f"{n}.{k}": p.grad.detach().clone() for n in ("sdf", "color", "refcolor", "nerf") if nets[n] is not None
                     for k, p in nets[n].named_parameters()}
    assert any(k.startswith("refcolor.") for k in res["1"]) and any(k.startswith("color.") for k in res["1"])
    for k in res["1"]:
        scale = res["0"][k].abs().max().item() + 1e-12
        assert (res["1"][k] - res["0"][k]).abs().max().item() <= 2e-5 * scale + 1e-9, (k, scale)


@pytest.mark.parametrize("rows", [1, 40])
def test_constant_background_colour_inside_the_compositing_kernels(rows):
    """renderer.py:367-368 `color + background_rgb * (1 - weights_sum)` in fneus_composite_fwd / _bwd (back_rgb) against the two
    element-wise lines behind the kernel: same colour bit for bit, same gradients of sdf / normal / rgb"""
    from fneus import synth
    from fneus.autograd import CompositeFn
    B, n = 40, 64
    g = torch.Generator().manual_seed(2)
    data = torch.from_numpy(synth.ray_batch(B, seed=3)).to(DEV)
    ro, rd = data[:, :3].contiguous(), data[:, 3:6].contiguous()
    mid = -(ro * rd).sum(-1, keepdim=True)
    z = (mid - 1.0 + 2.0 * torch.rand(B, n, generator=g).sort(dim=1).values.to(DEV)).contiguous()
    dists = torch.full_like(z, 2.0 / n)
    var = torch.tensor(0.3, device=DEV, requires_grad=True)
    back = torch.rand(rows, 3, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        sdf = (torch.randn(B * n, generator=torch.Generator().manual_seed(5)) * 0.2).to(DEV).requires_grad_(True)
        nrm = torch.nn.functional.normalize(torch.randn(B * n, 3, generator=torch.Generator().manual_seed(6)), dim=-1).to(DEV).requires_grad_(True)
        rgb = torch.rand(B * n, 3, generator=torch.Generator().manual_seed(7)).to(DEV).requires_grad_(True)
        out = CompositeFn.apply(sdf, nrm, rgb, var, ro, rd, z, dists, 0.5, None, None, back if fused else None)
        color, wsum = out[0], out[2]
        if not fused:
            color = color + back * (1.0 - wsum[:, None])
        w = torch.rand(B, 3, generator=torch.Generator().manual_seed(8)).to(DEV)
        ((color * w).sum() + 0.3 * wsum.sum()).backward()
        res.append((color.detach(), sdf.grad, nrm.grad, rgb.grad))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item() + 1e-12
