"""TEST HELPER (not product code): the stage-1 training losses (reference exp_runner.py:141-177) as plain torch ops on a render
result dict, written without boolean indexing: x[sdf_mask].sum() == (x * sdf_mask).sum().  The product computes them in
fneus_stage1_loss (csrc/loss_kernels.hip); the tests differentiate this formulation through the HIP render to check it."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def stage1_loss(render_out: dict, true_rgb, mask_in, igr_weight: float, mask_weight: float, surface_weight: float,
                mask_sum_global=None, mask_sdf_sum_global=None):
    if mask_weight > 0.0:
        mask = (mask_in > 0.5).float()
    else:
        mask = torch.ones_like(mask_in)
    mask_sum = mask.sum() + 1e-5 if mask_sum_global is None else mask_sum_global
    color_fine = render_out["color_fine"]
    color_error = (color_fine - true_rgb) * mask
    color_fine_loss = color_error.abs().sum() / mask_sum
    sm = render_out["sdf_mask"].float()[:, None]
    mask_sdf_sum = (mask * sm).sum() + 1e-5 if mask_sdf_sum_global is None else mask_sdf_sum_global
    surface_err = surface_weight * (render_out["surface_color"] - true_rgb) * mask * sm
    surface_color_loss = surface_err.abs().sum() / mask_sdf_sum
    eikonal_loss = render_out["gradient_error"]
    mask_loss = F.binary_cross_entropy(render_out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    loss = color_fine_loss + surface_color_loss + eikonal_loss * igr_weight + mask_loss * mask_weight
    with torch.no_grad():
        psnr = 20.0 * torch.log10(1.0 / (((color_fine - true_rgb) ** 2 * mask).sum() / (mask_sum * 3.0)).sqrt())
    return {"loss": loss, "color_loss": color_fine_loss, "surface_loss": surface_color_loss,
            "eikonal_loss": eikonal_loss, "mask_loss": mask_loss, "psnr": psnr}
