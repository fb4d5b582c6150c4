"""GPU parity of the fused per-ray tail (csrc/loss_kernels.hip): surface gather, and shading + blend + stage-1 losses
with their gradients, against the same maths written with torch ops (itself pinned to the oracle and the reference's
golden losses in tests/test_host_cpu.py / tests/test_hip_render.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _case(B, seed, mask_weight):
    rs = np.random.RandomState(seed)
    f = lambda *s: torch.from_numpy(rs.uniform(0.0, 1.0, size=s).astype(np.float32)).to(DEV)
    color, true_rgb = f(B, 3), f(B, 3)
    mask = (f(B, 1) > 0.3).float() * 0.9 + 0.05          # values around the 0.5 threshold on both sides
    wsum = f(B)
    wsum[:4] = torch.tensor([0.0, 5e-4, 0.9995, 1.0], device=DEV)      # outside the BCE clip range: zero gradient
    eik_num, eik_den = f(B) * 3.0, f(B) + 0.5
    diffuse, spec = f(2 * B, 3), f(2 * B, 3)
    diffuse[:3] *= 1e-3                                   # linear branch of the sRGB curve
    spec[:3] *= 1e-3
    diffuse[3:6] = 0.8                                    # brdf > 1 after the transfer: clipped, zero gradient
    spec[3:6] = 0.9
    wpair = f(B, 2) * 0.3
    sdf_mask = (f(B) > 0.25).to(torch.uint8)
    return dict(color=color, true_rgb=true_rgb, mask=mask, wsum=wsum, eik_num=eik_num, eik_den=eik_den, diffuse=diffuse,
                spec=spec, wpair=wpair, sdf_mask=sdf_mask, mask_weight=mask_weight)


def _torch_formulation(c, igr=0.1, sw=0.1):
    from _helper_losses import stage1_loss
    from models.fields import RefColor
    B = c["color"].shape[0]
    leaves = {k: c[k].clone().requires_grad_(True) for k in ("color", "wsum", "eik_num", "wpair", "diffuse", "spec")}
    ref = RefColor.shade(leaves["diffuse"], leaves["spec"])
    sm = c["sdf_mask"].bool()
    w_lo, w_hi = leaves["wpair"][:, 0:1] + 1e-5, leaves["wpair"][:, 1:2] + 1e-5
    ones = torch.ones(B, 3, device=DEV)

    def blend(v):
        v = v.reshape(B, 2, 3)
        return torch.where(sm[:, None], (v[:, 0] * w_lo + v[:, 1] * w_hi) / (w_lo + w_hi), ones)

    out = {"color_fine": leaves["color"], "surface_color": blend(ref["rgb"]), "sdf_mask": sm,
           "gradient_error": leaves["eik_num"].sum() / (c["eik_den"].sum() + 1e-5), "weight_sum": leaves["wsum"][:, None]}
    losses = stage1_loss(out, c["true_rgb"], c["mask"], igr, c["mask_weight"], sw)
    losses["loss"].backward()
    return losses, out, {k: v.grad for k, v in leaves.items()}, blend(ref["specular_rgb"]), blend(ref["diffuse_rgb"])


@pytest.mark.parametrize("B,mask_weight", [(512, 0.1), (37, 0.1), (1500, 0.0)])
def test_fused_loss_matches_torch_formulation(B, mask_weight):
    from fneus.autograd import Stage1LossFn
    c = _case(B, 7 + B, mask_weight)
    ref_losses, ref_out, ref_grads, ref_spec, ref_diff = _torch_formulation(c)
    leaves = {k: c[k].clone().requires_grad_(True) for k in ("color", "wsum", "eik_num", "wpair", "diffuse", "spec")}
    loss, lvec, surf, specc, diffc = Stage1LossFn.apply(leaves["color"], leaves["wsum"], leaves["eik_num"], leaves["wpair"],
                                                       leaves["diffuse"], leaves["spec"], c["eik_den"], c["true_rgb"],
                                                       c["mask"], c["sdf_mask"], 0.1, mask_weight, 0.1)
    (2.0 * loss).backward()                 # a non-unit cotangent must scale every gradient
    names = ["loss", "color_loss", "surface_loss", "eikonal_loss", "mask_loss", "psnr"]
    for i, k in enumerate(names):
        a, b = float(lvec[i]), float(ref_losses[k].detach())
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (k, a, b)
    assert abs(float(loss.detach()) - float(ref_losses["loss"].detach())) <= 2e-5
    assert (surf - ref_out["surface_color"]).abs().max().item() <= 1e-5
    assert (specc - ref_spec).abs().max().item() <= 1e-5
    assert (diffc - ref_diff).abs().max().item() <= 1e-5
    for k, g in ref_grads.items():
        got = leaves[k].grad / 2.0
        if k == "spec":       # only column 0 carries the specular value; torch spreads it over three via repeat()
            assert got[:, 1:].abs().max().item() == 0.0
        scale = g.abs().max().item() + 1e-12
        err = (got - g).abs().max().item() / scale
        print(f"  B={B} d{k}: rel max err {err:.2e} (scale {scale:.2e})")
        assert err <= 2e-4, (k, err)


def test_surface_gather_matches_indexing():
    from fneus import ops
    rs = np.random.RandomState(3)
    B, n = 77, 24
    feat = torch.from_numpy(rs.standard_normal((B * n, 256)).astype(np.float32)).to(DEV)
    normal = torch.from_numpy(rs.standard_normal((B * n, 3)).astype(np.float32)).to(DEV)
    mid_z = torch.from_numpy(rs.uniform(0, 2, size=(B, n)).astype(np.float32)).to(DEV)
    min_idx = torch.from_numpy(rs.randint(1, n, size=B).astype(np.int32)).to(DEV)
    sdf_mask = torch.from_numpy((rs.uniform(size=B) > 0.3).astype(np.uint8)).to(DEV)
    sel, t_sel, fs, ns = ops.surface_gather(min_idx, sdf_mask, mid_z, feat, normal)
    hi = torch.where(sdf_mask.bool(), min_idx.long(), torch.ones_like(min_idx, dtype=torch.long))   # renderer.py:316-321
    rows = torch.arange(B, device=DEV) * n
    want = torch.stack([rows + hi - 1, rows + hi], dim=1).reshape(-1)
    assert torch.equal(sel.long(), want)
    assert torch.equal(fs, feat[want]) and torch.equal(ns, normal[want]) and torch.equal(t_sel, mid_z.reshape(-1)[want])
