"""ctypes binding of libfneus_hip.so (C ABI declared in include/fneus.h).

The product path has no CPU / PyTorch fallback: if the shared library is missing or a symbol cannot be
resolved, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

# Import torch BEFORE loading the library: both link libamdhip64, and the process must end up with ONE HIP runtime
# (the one PyTorch ships).  Loading ours first binds it to /opt/rocm's copy and the two runtimes then disagree about
# devices ("no ROCm-capable device is detected").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# FNEUS_LIB selects another build of the same library (kernel experiments, tools/experiments/build_variant.sh)
LIB_PATH = os.environ.get("FNEUS_LIB") or os.path.join(_HERE, "libfneus_hip.so")


class FneusSdfStash(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("pe_hi", "pe_lo", "h_hi", "h_lo", "a_hi", "a_lo", "feat_hi", "feat_lo", "ps", "qs")]


class FneusSdfBwdBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("qbar_hi", "qbar_lo", "adj_hi", "adj_lo", "zbar_hi", "zbar_lo", "zsdf_hi", "zsdf_lo", "c_hi", "c_lo")]


class FneusColStash(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("side_hi", "side_lo", "u_hi", "u_lo", "zbar_hi", "zbar_lo", "zout_hi", "zout_lo", "mask", "feat_hi", "feat_lo", "dfeat_hi")] + [
                    ("dnormal_add", C.c_int32)]


class FneusNerfStash(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("pe_hi", "pe_lo", "h_hi", "h_lo", "feat_hi", "feat_lo", "dpe_hi", "dpe_lo", "hv_hi", "hv_lo", "mask",
                 "zbar_hi", "zbar_lo", "zfeat_hi", "zfeat_lo", "zhv_hi", "zhv_lo", "zout_hi", "zout_lo")]


class FneusSamplerStep(C.Structure):
    _fields_ = [("z_old", C.c_void_p), ("s_old", C.c_void_p), ("m", C.c_int), ("z_new", C.c_void_p), ("inv_s", C.c_float),
                ("z_out", C.c_void_p), ("s_out", C.c_void_p), ("z_next", C.c_void_p)]


class FneusAdamSegment(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("count", C.c_long)]


class FneusPackTask(C.Structure):
    _fields_ = [("jobs", C.c_void_p), ("n_jobs", C.c_int), ("n_units", C.c_int), ("maps", C.c_void_p), ("params", C.c_void_p),
                ("rowscale", C.c_void_p), ("invnorm", C.c_void_p), ("blob", C.c_void_p), ("rows", C.c_void_p), ("n_rows", C.c_int)]


class FneusWnTask(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("n_rows", C.c_int), ("bias_segs", C.c_void_p), ("n_segs", C.c_int), ("raw", C.c_void_p),
                ("rowscale", C.c_void_p), ("invnorm", C.c_void_p), ("d_eff", C.c_void_p), ("d_raw", C.c_void_p)]


class FneusMlpJob(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "weight", "bias", "y", "dy", "dx", "d_weight", "d_bias")] + \
               [(n, C.c_int) for n in ("rows", "n_in", "n_out", "act", "act_in")]


class FneusGemmPPJob(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("a_hi", "a_lo", "b_hi", "b_lo", "a2_hi", "a2_lo", "b2_hi", "b2_lo")] + \
               [(n, C.c_uint32) for n in ("a_blk", "b_blk", "a2_blk", "b2_blk")] + \
               [(n, C.c_uint16) for n in ("a_f0", "b_f0", "a2_f0", "b2_f0")] + \
               [("mt", C.c_int32), ("nt", C.c_int32), ("c", C.c_void_p), ("bias", C.c_void_p),
                ("ldc", C.c_int32), ("m", C.c_int32), ("n", C.c_int32), ("scale", C.c_float),
                ("wg_base", C.c_int32), ("splits", C.c_int32), ("n_tiles", C.c_int32), ("pad_", C.c_int32), ("n_dev", C.c_void_p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C factored-neus_amd/csrc` (or __graft_entry__.build()). "
            "The fneus product path has no fallback implementation.")
    lib = C.CDLL(LIB_PATH)
    vp, ip, f, l = C.c_void_p, C.c_int, C.c_float, C.c_long
    sigs = {
        "fneus_version": (C.c_int, []),
        "fneus_last_error": (C.c_char_p, []),
        "fneus_layout": (C.c_int, [ip, vp, ip]),
        "fneus_pack": (C.c_int, [vp, ip, ip, vp, vp, vp, vp, vp]),
        "fneus_rowscale": (C.c_int, [vp, ip, vp, vp, vp, vp]),
        "fneus_refresh_multi": (C.c_int, [C.POINTER(FneusPackTask), ip, vp]),
        "fneus_wn_backward_multi": (C.c_int, [C.POINTER(FneusWnTask), ip, vp]),
        "fneus_wn_backward": (C.c_int, [vp, ip, vp, ip, vp, vp, vp, vp, vp, vp]),
        "fneus_sdf_fwd": (C.c_int, [vp, vp, vp, vp, vp, ip, l, vp, ip, vp]),
        "fneus_sdf_fwd_rays": (C.c_int, [vp, vp, vp, vp, ip, l, vp, f, vp, vp, ip, vp]),
        "fneus_sdf_fwd_grad": (C.c_int, [vp, vp, vp, vp, vp, ip, l, C.POINTER(FneusSdfStash), vp, vp, vp, ip, ip, vp]),
    }
    optional = {
        "fneus_sdf_bwd": (C.c_int, [vp, vp, vp, vp, vp, ip, l, C.POINTER(FneusSdfStash), C.POINTER(FneusSdfBwdBufs),
                                    vp, vp, vp, ip, vp]),
        "fneus_color_fwd": (C.c_int, [vp, vp, vp, vp, vp, ip, l, vp, vp, vp, C.POINTER(FneusColStash), vp, ip, ip, vp]),
        "fneus_color_bwd": (C.c_int, [vp, l, vp, vp, C.POINTER(FneusColStash), vp, vp, ip, vp]),
        "fneus_color_out_dw": (C.c_int, [vp, vp, vp, vp, l, vp, vp, vp, vp, ip, vp, vp]),
        "fneus_color_out_dw_scratch_floats": (C.c_int, []),
        "fneus_probe_mfma": (C.c_int, [vp, ip, vp, vp, vp]),
        "fneus_probe_mfma_flops": (C.c_long, [ip]),
        "fneus_probe_copy": (C.c_int, [vp, vp, l, vp]),
        "fneus_refcolor_fwd": (C.c_int, [vp, ip, vp, vp, vp, vp, ip, l, vp, vp, vp, C.POINTER(FneusColStash), vp, ip, ip, vp]),
        "fneus_refcolor_bwd": (C.c_int, [vp, ip, l, vp, ip, vp, vp, vp, vp, C.POINTER(FneusColStash), vp, vp, ip, vp]),
        "fneus_refcolor_fwd_both": (C.c_int, [vp, vp, vp, vp, vp, vp, ip, l, vp, vp, vp, C.POINTER(FneusColStash),
                                              C.POINTER(FneusColStash), vp, vp, ip, ip, vp]),
        "fneus_refcolor_bwd_both": (C.c_int, [vp, vp, l, vp, ip, vp, vp, vp, vp, vp, vp, C.POINTER(FneusColStash),
                                              C.POINTER(FneusColStash), vp, vp, ip, vp]),
        "fneus_dw_gemm_pp": (C.c_int, [vp, ip, ip, l, vp, ip, vp]),
        "fneus_dw_gemm_pp_det": (C.c_int, [vp, ip, ip, l, vp, ip, vp, l, vp]),
        "fneus_nerf_bg_fwd": (C.c_int, [vp, vp, vp, l, C.POINTER(FneusNerfStash), vp, vp, ip, ip, vp, vp]),
        "fneus_nerf_bg_bwd": (C.c_int, [vp, l, vp, vp, C.POINTER(FneusNerfStash), ip, vp, vp]),
        "fneus_adam": (C.c_int, [C.POINTER(FneusAdamSegment), ip, vp, vp, C.c_double, C.c_double, C.c_double, ip, vp]),
        "fneus_surface_gather": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, ip, ip, vp, vp, vp, vp, vp, vp]),
        "fneus_surface_scatter": (C.c_int, [vp, vp, vp, ip, C.c_long, vp, vp, vp]),
        "fneus_surface_scatter_plane": (C.c_int, [vp, vp, vp, ip, C.c_long, vp, C.c_long, vp, vp]),
        "fneus_stage1_loss": (C.c_int, [vp] * 11 + [ip, f, f, f] + [vp] * 10 + [vp, vp]),
        "fneus_stage1_norms": (C.c_int, [vp, vp, vp, ip, f, vp, vp]),
        "fneus_upsample": (C.c_int, [vp, vp, vp, vp, ip, ip, ip, f, vp, vp]),
        "fneus_merge": (C.c_int, [vp, vp, ip, vp, vp, ip, ip, vp, vp, vp]),
        "fneus_merge_upsample": (C.c_int, [vp, vp, vp, vp, ip, vp, vp, ip, ip, f, ip, vp, vp, vp, vp, f, vp, vp, vp]),
        "fneus_sdf_fwd_merge_upsample": (C.c_int, [vp, vp, vp, vp, vp, ip, vp, ip, ip, f, ip, vp, vp, vp, vp, f, vp, vp, vp, ip, vp]),
        "fneus_sdf_fwd_merge_upsample_steps": (C.c_int, [vp, vp, vp, ip, C.POINTER(FneusSamplerStep), ip, ip, ip, vp, f, vp, vp, ip, vp]),
        "fneus_sections": (C.c_int, [vp, ip, ip, f, vp, vp, vp]),
        "fneus_ray_setup": (C.c_int, [vp, vp, vp, vp, vp, ip, ip, vp, vp]),
        "fneus_split_batch": (C.c_int, [vp, ip, vp, vp, vp, vp, vp]),
        "fneus_gen_random_rays": (C.c_int, [vp, vp, vp, vp, ip, ip, vp, vp, ip, vp, vp]),
        "fneus_gen_rays_grid": (C.c_int, [vp, vp, vp, vp, ip, ip, vp, vp, vp]),
        "fneus_outside_points": (C.c_int, [vp, vp, vp, ip, ip, f, vp, vp, vp, vp]),
        "fneus_outside_z": (C.c_int, [vp, vp, vp, vp, ip, ip, ip, vp, vp]),
        "fneus_vis_sample_dirs": (C.c_int, [vp, vp, vp, vp, ip, ip, vp, vp, vp]),
        "fneus_indir_sgs": (C.c_int, [vp, l, vp, vp]),
        "fneus_srgb_fwd": (C.c_int, [vp, l, ip, vp, vp]),
        "fneus_srgb_bwd": (C.c_int, [vp, vp, l, ip, vp, vp]),
        "fneus_indir_illum_fwd": (C.c_int, [vp, vp, ip, ip, ip, vp, vp]),
        "fneus_indir_illum_bwd": (C.c_int, [vp, vp, vp, ip, ip, ip, vp, vp]),
        "fneus_stage2_loss": (C.c_int, [vp, vp, vp, vp, vp, ip, vp, vp, vp, vp]),
        "fneus_stage3_loss": (C.c_int, [vp, vp, vp, vp, ip, vp, vp, vp]),
        "fneus_latent_kl_fwd": (C.c_int, [vp, vp, ip, f, ip, vp, vp]),
        "fneus_latent_kl_bwd": (C.c_int, [vp, vp, ip, f, ip, vp, vp, vp, vp]),
        "fneus_vis_sample_dirs_sgs": (C.c_int, [vp, vp, vp, ip, ip, vp, vp, vp]),
        "fneus_material_inputs": (C.c_int, [vp, vp, vp, ip, vp, vp, vp, vp, vp]),
        "fneus_sg_combine_fwd": (C.c_int, [vp, l, ip, vp, vp]),
        "fneus_sg_combine_bwd": (C.c_int, [vp, vp, l, ip, vp, vp]),
        "fneus_outside_select": (C.c_int, [vp, vp, vp, vp, ip, ip, ip, f, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "fneus_outside_alpha_sel_fwd": (C.c_int, [vp, vp, vp, vp, vp, l, vp, vp, vp]),
        "fneus_outside_alpha_sel_bwd": (C.c_int, [vp, vp, vp, vp, vp, l, vp, vp, vp, vp, vp]),
        "fneus_outside_alpha_fwd": (C.c_int, [vp, vp, vp, l, vp, vp, vp]),
        "fneus_outside_alpha_bwd": (C.c_int, [vp, vp, vp, vp, vp, l, vp, vp, vp]),
        "fneus_sg_render_fwd": (C.c_int, [vp] * 6 + [ip, ip, ip, f, vp, vp]),
        "fneus_sg_render_bwd": (C.c_int, [vp] * 6 + [ip, ip, ip, f, vp, vp, vp, vp]),
        "fneus_sg_render_heads_fwd": (C.c_int, [vp] * 7 + [ip, ip, ip, f, vp, vp]),
        "fneus_sg_render_heads_bwd": (C.c_int, [vp] * 7 + [ip, ip, ip, f, vp, vp, vp, vp, vp]),
        "fneus_embed": (C.c_int, [vp, l, ip, ip, vp, vp]),
        "fneus_mlp_forward": (C.c_int, [C.POINTER(FneusMlpJob), ip, vp]),
        "fneus_mlp_backward_input": (C.c_int, [C.POINTER(FneusMlpJob), ip, vp]),
        "fneus_mlp_backward_params": (C.c_int, [C.POINTER(FneusMlpJob), ip, vp]),
        "fneus_lvis_visibility": (C.c_int, [vp, vp, vp, vp, vp, vp, ip, ip, ip, vp, ip, vp]),
        "fneus_lvis_blob_bytes": (C.c_size_t, []),
        "fneus_lvis_h16_pack": (C.c_int, [vp, vp, vp]),
        "fneus_ray_hit": (C.c_int, [vp] * 7 + [ip, ip, f] + [vp] * 5 + [vp]),
        "fneus_sample_dirs": (C.c_int, [vp] * 4 + [ip, ip, vp, vp, vp]),
        "fneus_composite_fwd": (C.c_int, [vp] * 8 + [ip, ip, ip, f, vp, vp, vp, ip] + [vp] * 10 + [vp, ip, vp]),
        "fneus_composite_bwd": (C.c_int, [vp] * 8 + [ip, ip, ip, f, vp, vp, vp, ip] + [vp] * 13 + [vp, ip, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in optional.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def check(rc: int, what: str):
    if rc != 0:
        msg = lib.fneus_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
