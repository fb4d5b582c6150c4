#!/usr/bin/env python3
"""Benchmark of the Factored-NeuS stage-1 training hot path on MI355X (HIP backend).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--prec parity|fast] [--no-cpu-baseline] [--no-profile]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(one rank per GPU, RCCL; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

A "step" = one pass of the hot path over one synthetic DTU-shaped batch that is already resident in HBM:
render (hierarchical sampler, SDF + normal, colour, compositing) -> 4-term loss -> backward -> [gradient all-reduce]
-> Adam, for BASELINE.json configs[1]: 512 rays x (64+64) samples, confs/wmask.conf networks, random-init weights of
the reference's distributions.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))

import numpy as np
import torch
import torch.distributed as dist

RAYS, N_SAMPLES, N_IMPORTANCE = 512, 64, 64
SAMPLES_PER_STEP = RAYS * (N_SAMPLES + N_IMPORTANCE)

# algorithmic work, SURVEY.md section 8(d): GEMM MACs per ray-sample (1 MAC = 2 FLOP)
F_SDF = 524544 * 2.0       # one SDF-network forward
F_COL = 271360 * 2.0       # one colour-network forward
FLOP_TRAIN_PER_SAMPLE = (0.875 + 2 + 4) * F_SDF + 3 * F_COL      # = 8.84 MFLOP
# per-kernel algorithmic FLOPs per launch at N = 65 536 samples (used for the dominant-kernel roofline)
KERNEL_FLOPS = {
    "fneus_sdf_fwd_grad": 2 * F_SDF * SAMPLES_PER_STEP,           # value+feature forward and the reverse sweep
    "fneus_sdf_bwd": 2 * F_SDF * SAMPLES_PER_STEP,                # ascending + descending chains
    "fneus_dw_gemm_pp:sdf": 2 * F_SDF * SAMPLES_PER_STEP,         # dW = zbar^T u + a^T adj for the 9 SDF layers
    "fneus_dw_gemm_pp:color": F_COL * SAMPLES_PER_STEP,
    "fneus_color_fwd": F_COL * SAMPLES_PER_STEP,
    "fneus_color_bwd": F_COL * SAMPLES_PER_STEP,
}
# (single-GPU steps run the colour network's products inside the SDF network's launch: fneus/autograd.py ColorFn.backward)
KERNEL_FLOPS["fneus_dw_gemm_pp:sdf+color"] = KERNEL_FLOPS["fneus_dw_gemm_pp:sdf"] + KERNEL_FLOPS["fneus_dw_gemm_pp:color"]
PEAK_BF16_MFMA_TFLOPS = 2500.0       # MI355X dense bf16 (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0                # HBM3E (MI355X_MICROARCH.md)
# Algorithmic HBM bytes per sample of the fragment-plane design in parity mode with bf16 gradient planes (the default,
# gprec 1): what a launch must move even with perfect caching.  One 256-wide bf16 plane = 512 B per
# sample and layer, the sigma' plane (u16 fixed point) likewise.
#   K2 writes  h 8 x 512, a 8 x 512, sigma' 8 x 512, feature planes hi + lo 2 x 512 (round 6: no fp32 feature rows), PE plane 128, sdf + normal 16;
#      reads   sigma' back once for its reverse sweep 8 x 512, 12 B of ray data
#   K3 reads   sigma' twice (ascending + descending chain) 2 x 8 x 512, a 8 x 512, the coupling planes it wrote 8 x 512,
#              d_feat as the bf16 fragments the colour backward wrote 512 (round 6; fp32 rows before: 1024), cotangents 16;
#      writes  the coupling planes 8 x 512, adj 8 x 512, zbar 8 x 512 (slot 8 IS d_feat), qbar 128
KERNEL_BYTES_PARITY = {
    "fneus_sdf_fwd_grad": (3 * 8 * 512 + 2 * 512 + 128 + 16 + 8 * 512 + 12) * SAMPLES_PER_STEP,
    "fneus_sdf_bwd": (2 * 8 * 512 + 8 * 512 + 8 * 512 + 512 + 16 + 8 * 512 + 8 * 512 + 8 * 512 + 128) * SAMPLES_PER_STEP,
    # four bf16 planes per layer (h, a, zbar, adj) + PE plane, read once
    "fneus_dw_gemm_pp:sdf": (4 * 8 * 512 + 512 + 128) * SAMPLES_PER_STEP,
}


# The colour network's design bytes per sample (gradient precision 2, the default): forward writes u 4 x 512 + the lo plane of u_3 512 +
# the side plane 128 + the masks 64 and reads the feature planes hi + lo 1024 + the normal rows 12; backward reads masks + d_rgb and writes
# zbar 4 x 512 + zout + d_feat as bf16 fragments 512 + d_normal 12;
# its GEMM reads u, zbar (8 x 512), the feature and side planes (512 + 128).
KERNEL_BYTES_PARITY_EXTRA = {
    "fneus_color_fwd": (4 * 512 + 512 + 128 + 64 + 1024 + 12 + 12) * SAMPLES_PER_STEP,
    "fneus_color_bwd": (64 + 12 + 4 * 512 + 64 + 512 + 12) * SAMPLES_PER_STEP,
    "fneus_dw_gemm_pp:color": (8 * 512 + 512 + 128 + 64) * SAMPLES_PER_STEP,
    "fneus_sdf_fwd": 0.875 * 16 * SAMPLES_PER_STEP,
    # gradient precision 2 (the default): the output layer's product reads the hi + lo planes of u_3 and d_rgb, rgb
    "fneus_color_out_dw": (2 * 512 + 24) * SAMPLES_PER_STEP,
}
KERNEL_BYTES_PARITY["fneus_dw_gemm_pp:sdf+color"] = (KERNEL_BYTES_PARITY["fneus_dw_gemm_pp:sdf"]
                                                     + KERNEL_BYTES_PARITY_EXTRA["fneus_dw_gemm_pp:color"])
ALGORITHMIC_STASH_BYTES_PER_SAMPLE = 11.0 * 1024 + 40      # ~11 KB activation stash + 40 B of API outputs
HBM_ACHIEVABLE_GBS = 6300.0          # float4 copy on this part (MI355X_MICROARCH.md: 6.29 TB/s measured, 79 % of the 8 TB/s spec)
# gradient precision 1 / 2: the weight-gradient GEMM multiplies bf16 planes (1 MFMA per product); the cotangent chains (K3, the colour
# network's backward) run on the bf16 values of those planes against W hi + lo (2 per product, round 6: DESIGN.md 4.1e); the forward
# chains and K2's reverse sweep issue 3
MFMAS_PER_PRODUCT = {"fneus_dw_gemm_pp:sdf": 1, "fneus_dw_gemm_pp:color": 1, "fneus_dw_gemm_pp:sdf+color": 1, "fneus_sdf_bwd": 2, "fneus_color_bwd": 2}
KERNEL_FLOPS_EXTRA = {"fneus_sdf_fwd": 0.875 * F_SDF * SAMPLES_PER_STEP}


def kernel_floor_ms(name, gprec=1):
    """floor of one step's launches of a kernel in the shipped design: max(design bytes / achievable HBM rate, MFMAs issued x
    algorithmic FLOPs / the dense bf16 peak).  The parity arithmetic issues 3 MFMAs per product in the chains, the
    weight-gradient GEMM 1 (gprec 1) or 3 (gprec 3)."""
    flops = KERNEL_FLOPS.get(name, KERNEL_FLOPS_EXTRA.get(name, 0.0))
    byts = KERNEL_BYTES_PARITY.get(name, KERNEL_BYTES_PARITY_EXTRA.get(name, 0.0))
    k = MFMAS_PER_PRODUCT.get(name, 3) if gprec in (1, 2) else 3
    t_mfma = k * flops / (PEAK_BF16_MFMA_TFLOPS * 1e12) * 1e3
    t_hbm = byts / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3
    return {"floor_ms": max(t_mfma, t_hbm), "mfma_ms": t_mfma, "hbm_ms": t_hbm, "bound": "mfma" if t_mfma >= t_hbm else "hbm"}


CPU_CONFIGS = {       # BASELINE.json configs[0] and configs[1]
    "cfg1": dict(rays=256, n_samples=32, n_importance=32),
    "cfg2": dict(rays=RAYS, n_samples=N_SAMPLES, n_importance=N_IMPORTANCE),
}


def _cpu_model():
    try:
        return next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        return ""


def oracle_train_steps(cfg: dict, min_steps: int, budget_s: float, device=None):
    """`min_steps` (or more while the budget lasts) FULL-SIZE train steps of the CPU port of the reference algorithm
    (oracle/ref_torch.py: render -> 4-term loss -> autograd backward -> Adam) after one warm-up step -> per-step seconds"""
    from oracle import ref_torch as R
    from fneus import synth
    on_gpu = device is not None
    dev = device if on_gpu else torch.device("cpu")
    n_rays, n_s, n_i = cfg["rays"], cfg["n_samples"], cfg["n_importance"]
    T = lambda sd: {k: torch.from_numpy(v).clone().to(dev).requires_grad_(True) for k, v in sd.items()}
    sd_sdf, sd_col, sd_ref = T(synth.sdf_state_dict(0)), T(synth.color_state_dict(1)), T(synth.refcolor_state_dict(2))
    variance = torch.tensor(0.3, device=dev, requires_grad=True)
    leaves = list(sd_sdf.values()) + list(sd_col.values()) + list(sd_ref.values()) + [variance]
    opt = torch.optim.Adam(leaves, lr=5e-4)
    times = []
    t_start = time.time()
    step = 0
    while True:
        data = torch.from_numpy(synth.ray_batch(n_rays, seed=1000 + step)).to(dev)
        rays_o, rays_d, rgb, mask = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
        if on_gpu:
            torch.cuda.synchronize()
        t0 = time.time()
        near, far = R.near_far_from_sphere(rays_o, rays_d)
        out = R.render(rays_o, rays_d, near, far, R.sdf_params_from_state_dict(sd_sdf), R.inv_s_from_variance(variance),
                       R.color_params_from_state_dict(sd_col), sd_ref, None, n_samples=n_s, n_importance=n_i,
                       t_rand=torch.rand(n_rays, 1, device=dev), cos_anneal_ratio=1.0)
        losses = R.stage1_loss(out, rgb, mask, 0.1, 0.1, 0.1)
        opt.zero_grad()
        losses["loss"].backward()
        opt.step()
        if on_gpu:
            torch.cuda.synchronize()
        if step > 0:
            times.append(time.time() - t0)
        step += 1
        if len(times) >= min_steps and (time.time() - t_start > budget_s or len(times) >= 8):
            break
    return times


def cpu_baseline(device=None):
    """The oracle (CPU port of the reference algorithm) timed on the host cores at FULL size: configs[1] (512 rays x
    (64+64), the headline workload; `value`) and configs[0] (256 x (32+32), the reference's own CPU-runnable case), at
    least 3 timed steps each after one warm-up, median.  With `device` = the GPU the same port runs through stock
    PyTorch-ROCm ops: the "unfused GPU" figure the reference itself would get on this box."""
    if device is not None:
        times = oracle_train_steps(CPU_CONFIGS["cfg2"], 7, 0.0, device)
        t = float(np.median(times))
        return {"value": SAMPLES_PER_STEP / t, "unit": "ray-samples/s", "ms_per_step": t * 1e3,
                "sample": f"{len(times)} full {RAYS}-ray train steps of oracle/ref_torch.py (fp32, eager stock PyTorch-ROCm "
                          f"ops, autograd double backward) on the same GPU after 1 warm-up"}
    threads = min(32, os.cpu_count() or 1)     # eager torch on small ops gets slower, not faster, beyond ~32 threads
    torch.set_num_threads(threads)
    legs = {}
    for name, min_steps, budget in (("cfg2", 3, 20.0), ("cfg1", 3, 6.0)):
        cfg = CPU_CONFIGS[name]
        times = oracle_train_steps(cfg, min_steps, budget)
        t = float(np.median(times))
        n = cfg["rays"] * (cfg["n_samples"] + cfg["n_importance"])
        legs[name] = {"value": n / t, "unit": "ray-samples/s", "s_per_step": t, "steps": len(times), "cores": threads,
                      "workload": f"{cfg['rays']} rays x ({cfg['n_samples']}+{cfg['n_importance']}) samples, full train step"}
    c2 = legs["cfg2"]
    return {"value": c2["value"], "unit": "ray-samples/s", "cores": threads, "kind": "port", "cpu_model": _cpu_model(),
            "sample": f"{c2['steps']} full-size train steps (median {c2['s_per_step']:.2f} s/step) of {RAYS} rays x "
                      f"{N_SAMPLES + N_IMPORTANCE} samples (BASELINE configs[1], the headline workload) after 1 warm-up, "
                      f"oracle/ref_torch.py fp32, {threads} torch threads of {os.cpu_count()} host cores",
            "cfg1": legs["cfg1"], "cfg2": legs["cfg2"]}


# newest first: the PMC passes are re-collected whenever a kernel's memory behaviour changes (tools/collect_profiles.sh)
TRAFFIC_FILES = ("r06_t_traffic.json", "r06_u_traffic.json", "r06_v_traffic.json", "r06_w_traffic.json", "r06_z_traffic.json", "r06_y_traffic.json", "r06_x_traffic.json", "r05_e_traffic.json", "r05_d_traffic.json", "r05_c_traffic.json", "r05_b_traffic.json", "r05_a_traffic.json", "r04_h_traffic.json", "r04_f_traffic.json", "r04_e_traffic.json", "r04_d_traffic.json", "r04_c_traffic.json", "r04_b_traffic.json", "r04_a_traffic.json", "r03_h_traffic.json", "r03_g_traffic.json", "r03_f_traffic.json", "r03_e_traffic.json", "r03_d_traffic.json", "r03_c_traffic.json", "r03_b_traffic.json", "r03_a_traffic.json", "r02_e_traffic.json", "r02_d_traffic.json", "r02_c_traffic.json", "r02_b_traffic.json", "r02_a_traffic.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps after >= 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prec", choices=["parity", "fast"], default="parity")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-fast-extra", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step's kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--rays-global", type=int, default=0,
                    help="STRONG scaling: this many rays per step over all ranks (config 5 quotes 2048 = 256 per GPU at 8 ranks); "
                         "default 0 = weak scaling, 512 rays per rank")
    ap.add_argument("--womask", action="store_true", help="womask.conf shape (+ 32 background samples per ray through the NeRF++ kernels)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through the C stdio of the process, which is
    # flushed at exit -- AFTER Python's own buffer when stdout is a pipe or a file (seen on the MI355X box: five banner lines
    # behind the JSON line).  So: everything written to descriptor 1 during the run goes to stderr, and the JSON line is
    # written to the real stdout directly.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from fneus import ops
    from fneus.parallel import init_from_env, broadcast_parameters
    from fneus.trainer import Stage1Trainer, synthetic_batches

    # backend "nccl" = RCCL over xGMI.  FNEUS_DIST_BACKEND=gloo exists so that the N > 1 code path can be exercised with
    # several ranks sharing ONE GPU (RCCL refuses two ranks per device); it is not a benchmark configuration.
    rank, world, local = init_from_env(os.environ.get("FNEUS_DIST_BACKEND", "nccl"))
    local = local % max(torch.cuda.device_count(), 1)
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:       # one line per rank: a SCALE record can show that N ranks on N devices took part
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = "?"
        print(f"[bench rank {rank}/{world}] device cuda:{local} {torch.cuda.get_device_name(local)} backend "
              f"{dist.get_backend()} RCCL {ver} MASTER {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
              file=sys.stderr, flush=True)

    # FNEUS_DP_SINGLE=1: the data-parallel step (four graph segments around three RCCL collectives) with ONE rank -- what the
    # structure of the N > 1 step costs before any wire time; a diagnostic, not the N = 1 configuration
    dp_single = os.environ.get("FNEUS_DP_SINGLE", "0") == "1" and dist.is_initialized()

    strong = args.rays_global > 0
    rays_rank = max(32, args.rays_global // world) if strong else RAYS
    n_out = 32 if args.womask else 0
    samples_rank = rays_rank * (N_SAMPLES + N_IMPORTANCE + n_out)

    def run(prec, steps, warmup, profile, gprec=None):
        kw = {}
        if args.womask:
            import copy
            from fneus.trainer import WMASK_MODEL
            conf = copy.deepcopy(WMASK_MODEL)
            conf["neus_renderer"]["n_outside"] = 32
            kw["model_conf"] = conf
        tr = Stage1Trainer(device, prec=prec, distributed=(world > 1 or dp_single), use_graph=not args.no_graph, gprec=gprec, **kw)
        broadcast_parameters(tr.modules)
        batches = synthetic_batches(steps + warmup + (3 if profile else 0), rays_rank, device, rank=rank)
        step_kw = dict(cos_anneal_ratio=0.5, background_rgb=torch.ones(1, 3, device=device)) if args.womask else {}
        # more than one rank: the job measures both forms of its gradient exchange on its own steps (untimed, ahead of the W
        # warm-up steps) and keeps the faster one -- every rank takes the same decision (fneus/trainer.py autotune_exchange)
        if world > 1 or dp_single:
            choice = tr.autotune_exchange(batches[:max(4, min(len(batches), 8))], **step_kw)
            run.exchange = choice
            print(f"[bench rank {rank}/{world}] gradient exchange: {choice['choice']} (split {choice['ms_split']} ms, single "
                  f"{choice['ms_single']} ms per step over {choice['steps']} steps; {choice['why']})", file=sys.stderr, flush=True)
        _step = tr.train_step
        tr.train_step = lambda b: _step(b, **step_kw)
        for i in range(warmup):
            tr.train_step(batches[i])
        if tr.use_graph and not tr._graphs:      # too few warm-up steps to have captured: capture now, untimed
            for i in range(tr.graph_warmup_steps + 1):
                tr.train_step(batches[i % len(batches)])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            tr.train_step(batches[warmup + i])
            marks[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        run.step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        prof = None
        if profile:
            ops.profile_begin()
            for i in range(3):
                tr.train_step(batches[warmup + steps + i])
            prof = ops.profile_end()
        return dt, prof, tr

    prec = ops.PREC_PARITY if args.prec == "parity" else ops.PREC_FAST
    dt, prof, tr = run(prec, args.steps, args.warmup, profile=not args.no_profile)
    box = None
    if rank == 0:
        try:        # what THIS lease delivers to two trivial kernels (the same code measured 5-8 % apart between boxes)
            box = ops.box_probe(device)
            box["device"] = torch.cuda.get_device_name(local)
            box["reference"] = "builder's boxes, round 6: see DESIGN.md 5.0 (a box whose probes read lower runs every kernel of the step slower)"
        except Exception as e:
            box = {"error": repr(e)}
    ms_per_step = dt / args.steps * 1e3
    value = world * samples_rank * args.steps / dt
    gprec_run = ops.DEFAULT_GPREC if hasattr(ops, "DEFAULT_GPREC") else 1

    result = {
        "metric": "train-step ray-samples/s (stage-1, 512x128)",
        "value": value,
        "unit": "ray-samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_median": float(np.median(run.step_ms)),      # per-step HIP events on the launch stream (this rank)
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": (f"bf16x3 split MFMA (3 products per value, fp32 accumulate) in every chain whose result is a forward value (sdf, features, "
                  f"normals, colours: the 1e-4 parity mode); the cotangent chains of the backward (K3, colour backward) on W hi + lo x the bf16 "
                  f"cotangents their planes hold (2 products) unless the gradient precision is 3; "
                  f"weight-gradient GEMM operands: gradient precision {gprec_run} "
                  + {1: "(bf16 planes)", 2: "(bf16 planes; the colour network's output layer -- the one product whose bf16 rounding exceeds "
                                          "the exact mode's gradient bounds -- on exact operands, fp32 FMAs)", 3: "(bf16 hi + lo planes)"}[gprec_run])
                 if prec == 3 else "bf16 MFMA, fp32 accumulate",
        "data": "synthetic DTU-shaped rays (one camera per step), random-init weights of the reference distributions",
        "config": {"workload": (f"Shiny-Blender-shaped womask.conf stage-1 train step, {rays_rank * world} rays per step over {world} rank(s) x "
                                f"(64+64+32) samples (BASELINE configs[4] shape)" if args.womask else
                                f"dtu_scan97-shaped wmask.conf stage-1 SDF+radiance train step, {rays_rank} rays x (64+64) samples, "
                                "1xMI355X per rank"), "rays_per_gpu": rays_rank, "samples_per_ray": N_SAMPLES + N_IMPORTANCE + n_out,
                   **({"rays_global": rays_rank * world} if strong else {}),
                   "parallelism": (f"dp{world} (ray-sharded replicas; gradient arena all-reduced in place, "
                                   + (("in two parts, the early one beside the SDF backward" if tr.split_exchange else
                                       "in one exchange behind the backward") if (world > 1 or dp_single) else "no exchange with one rank")
                                   + (f"; chosen at start-up on this job's own steps: split {run.exchange['ms_split']} ms vs single "
                                      f"{run.exchange['ms_single']} ms per step" if getattr(run, "exchange", None) and run.exchange.get("steps") else "")
                                   + "; >1 rank over RCCL unmeasured on the builder's 1-GPU boxes)"),
                   "launch": ("eager kernel launches" if not (tr.use_graph and tr._graphs) else
                              ("four hipGraph replays per step around the three collectives (loss normalisers; early part of the gradient "
                               "arena beside the SDF backward; late part)" if tr.split_exchange else
                               "three hipGraph replays per step around the two collectives (loss normalisers; gradient arena)")
                              if (world > 1 or dp_single) else
                              "one hipGraph replay per step")},
        **({"diagnostic": "FNEUS_DP_SINGLE=1: data-parallel step structure with one rank"} if dp_single else {}),
        "mfma_roofline_frac_step": value / world * FLOP_TRAIN_PER_SAMPLE / (PEAK_BF16_MFMA_TFLOPS * 1e12),
        **({"box": box} if box is not None else {}),
    }
    standard = not strong and not args.womask          # the extras below describe the headline workload only

    if rank == 0 and prof and standard:
        per = {}
        for name, (n, ms) in prof.items():
            per[name] = {"launches_per_step": n / 3.0, "ms_per_step": ms / 3.0, "avg_ms": ms / n}
        result["kernels_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1]["ms_per_step"])}
        dom = max((k for k in per if k in KERNEL_FLOPS), key=lambda k: per[k]["ms_per_step"])
        flops_per_launch = KERNEL_FLOPS[dom] / max(round(per[dom]["launches_per_step"]), 1)
        achieved = flops_per_launch / (per[dom]["avg_ms"] * 1e-3) / 1e12
        traffic = traffic_src = None      # HBM bytes per launch from the committed PMC passes (profiles/*_traffic.json), parity mode only
        if prec == ops.PREC_PARITY:
            for tag in TRAFFIC_FILES:
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", tag)))
                    traffic = tj["kernels"][dom.split(":")[0]]["hbm_bytes_per_launch"]
                    traffic_src = {"file": "profiles/" + tag, "kernel": tj["kernels"][dom.split(":")[0]].get("kernel"),
                                   "step_bytes_per_ray_sample": tj.get("step_total", {}).get("bytes_per_ray_sample")}
                    break
                except Exception:
                    continue
        if prec == ops.PREC_PARITY and dom in KERNEL_BYTES_PARITY:
            # the same launch seen against the HBM roofline: these kernels move their whole activation stash
            gbs = KERNEL_BYTES_PARITY[dom] / max(round(per[dom]["launches_per_step"]), 1) / (per[dom]["avg_ms"] * 1e-3) / 1e9
            result["roofline_hbm"] = {"kernel": dom, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                      "note": "algorithmic stash bytes per launch (DESIGN.md 4.1) / launch duration; "
                                              "streaming torch kernels reach 4.0 (read) - 6.8 (write) TB/s on this part"}
        # both roofs for the three kernels that carry the step (an extra: `roofline` above stays the contract's object)
        both = {}
        for k in ("fneus_sdf_fwd_grad", "fneus_sdf_bwd", "fneus_dw_gemm_pp:sdf", "fneus_dw_gemm_pp:sdf+color"):
            if k not in per or k not in KERNEL_FLOPS:
                continue
            n_l = max(round(per[k]["launches_per_step"]), 1)
            sec = per[k]["avg_ms"] * 1e-3
            e = {"avg_launch_ms": round(per[k]["avg_ms"], 4),
                 "mfma_frac": KERNEL_FLOPS[k] / n_l / sec / 1e12 / PEAK_BF16_MFMA_TFLOPS}
            if prec == ops.PREC_PARITY and k in KERNEL_BYTES_PARITY:
                e["hbm_frac_algorithmic_bytes"] = KERNEL_BYTES_PARITY[k] / n_l / sec / 1e9 / PEAK_HBM_GBS
            if prec == ops.PREC_PARITY and ":" not in k:
                for tag in TRAFFIC_FILES:
                    try:
                        tj = json.load(open(os.path.join(ROOT, "profiles", tag)))
                        e["hbm_frac_pmc_bytes"] = tj["kernels"][k]["hbm_bytes_per_launch"] / sec / 1e9 / PEAK_HBM_GBS
                        e["pmc_file"] = "profiles/" + tag
                        break
                    except Exception:
                        continue
            both[k] = e
        # the design's own floor, kernel by kernel and for the step: how far each kernel is from what its
        # bytes and its 3-MFMA products allow, whatever the roofline fraction says
        if prec == ops.PREC_PARITY:
            floors, floor_sum, measured_sum = {}, 0.0, 0.0
            for k in list(KERNEL_FLOPS) + list(KERNEL_FLOPS_EXTRA):
                if k not in per:
                    continue
                f = kernel_floor_ms(k, gprec_run)
                floors[k] = {"ms_per_step": round(per[k]["ms_per_step"], 4), "floor_ms": round(f["floor_ms"], 4), "bound": f["bound"],
                             "frac_of_floor": round(f["floor_ms"] / per[k]["ms_per_step"], 3)}
                floor_sum += f["floor_ms"]
                measured_sum += per[k]["ms_per_step"]
                if k in both:
                    both[k]["floor_ms"] = round(f["floor_ms"], 4)
                    both[k]["frac_of_floor"] = floors[k]["frac_of_floor"]
            rest = ms_per_step - measured_sum                   # sampler, compositing, RefColor heads, loss, optimiser, PyTorch remainder
            result["floor"] = {"kernels": floors, "step_floor_ms": round(floor_sum, 4),
                               "step_floor_plus_unmodelled_rest_ms": round(floor_sum + max(rest, 0.0), 4),
                               "ms_per_step": round(ms_per_step, 4), "frac_of_floor": round(floor_sum / ms_per_step, 3),
                               "note": "floor of a kernel = max(design HBM bytes / 6.3 TB/s achievable, MFMAs issued x algorithmic FLOPs / "
                                       "2.5 PFLOP/s): what THIS design (fragment-plane stash, 3 bf16 MFMAs per product, a separate "
                                       "weight-gradient GEMM) could reach with every launch at its own roof and no latency-bound tail; "
                                       "north_star's 2e8 ray-samples/s (0.33 ms per step) lies below this floor in parity mode"}
        if both:
            result["rooflines_by_kernel"] = both
        # the contract's `roofline`: the dominant kernel against the roof that BINDS it in this design (its own `floor` entry: the larger
        # of design bytes / achievable HBM rate and MFMAs issued x algorithmic FLOPs / the dense peak), the other roof's fraction beside it
        fl = kernel_floor_ms(dom, gprec_run) if prec == ops.PREC_PARITY else {"bound": "mfma"}
        n_dom = max(round(per[dom]["launches_per_step"]), 1)
        mfma_frac = achieved / PEAK_BF16_MFMA_TFLOPS
        gbs_dom = (KERNEL_BYTES_PARITY.get(dom, KERNEL_BYTES_PARITY_EXTRA.get(dom, 0.0)) / n_dom / (per[dom]["avg_ms"] * 1e-3) / 1e9
                   if prec == ops.PREC_PARITY else 0.0)
        if fl["bound"] == "hbm" and gbs_dom > 0.0:
            result["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": gbs_dom, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": gbs_dom / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                  "avg_launch_ms": per[dom]["avg_ms"],
                                  "other_roof": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                 "frac": mfma_frac, "mfma_issue_frac": mfma_frac * (MFMAS_PER_PRODUCT.get(dom, 3) if gprec_run != 3 else 3)},
                                  "floor_ms": round(fl["floor_ms"], 4), "frac_of_floor": round(fl["floor_ms"] / per[dom]["avg_ms"] / n_dom, 3),
                                  "note": "the dominant kernel against the roof that binds it: algorithmic (design) stash bytes per launch / "
                                          "HIP-event launch duration against the 8 TB/s HBM3E peak (6.3 TB/s is what a copy reaches); "
                                          "`traffic` = PMC bytes per launch from the committed file named in traffic_source; other_roof: "
                                          "algorithmic FLOPs / duration against the dense bf16 MFMA peak (x 3 MFMAs issued per product in parity mode)"}
        else:
            result["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS,
                                  "unit": "TFLOP/s", "frac": mfma_frac, "traffic": traffic,
                                  "traffic_source": traffic_src, "avg_launch_ms": per[dom]["avg_ms"],
                                  "other_roof": ({"bound": "hbm", "achieved": gbs_dom, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs_dom / PEAK_HBM_GBS}
                                                 if gbs_dom > 0.0 else None),
                                  "note": "algorithmic (fp32-equivalent) FLOPs per launch / HIP-event launch duration; "
                                          "parity mode issues 3 bf16 MFMAs per algorithmic product"
                                          + ("; fneus_sdf_fwd_grad is two kernels since round 3 (forward chain with the stash, "
                                             "reverse sweep): FLOPs and duration of the pair" if dom == "fneus_sdf_fwd_grad" else "")}
        # HBM traffic of the whole step per ray sample (PMC, committed file) against SURVEY 8(d)'s algorithmic stash
        if prec == ops.PREC_PARITY and traffic_src and traffic_src.get("step_bytes_per_ray_sample"):
            result["bytes_per_ray_sample"] = traffic_src["step_bytes_per_ray_sample"]
            result["wasted_traffic_ratio"] = traffic_src["step_bytes_per_ray_sample"] / ALGORITHMIC_STASH_BYTES_PER_SAMPLE
            result["bytes_per_ray_sample_source"] = traffic_src["file"] + " (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE over one step, builder's box; / SURVEY 8(d)'s ~11 KB stash + 40 B)"

    if rank == 0 and world == 1 and not args.no_fast_extra and standard:
        # forward-only render of the same batch (what validate_image runs per ray chunk), SURVEY.md section 8(d)
        fb = synthetic_batches(4, RAYS, device, rank=rank)
        for b in fb[:2]:
            tr.render_only(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_fwd = 20
        for i in range(n_fwd):
            tr.render_only(fb[i % len(fb)])
        torch.cuda.synchronize()
        dt_r = (time.perf_counter() - t0) / n_fwd
        result["forward_only_render"] = {"value": SAMPLES_PER_STEP / dt_r, "unit": "ray-samples/s", "ms_per_call": dt_r * 1e3,
                                         "note": "NeuSRenderer.render under no_grad, no stash written; one hipGraph replay per chunk shape (round 5; FNEUS_RENDER_GRAPH=0: eager launches)"}

    if rank == 0 and world == 1 and not args.no_fast_extra and prec == ops.PREC_PARITY and standard:
        dt_f, _, _ = run(ops.PREC_FAST, max(args.steps // 2, 5), 3, profile=False)
        v = SAMPLES_PER_STEP * max(args.steps // 2, 5) / dt_f
        result["fast_bf16"] = {"value": v, "unit": "ray-samples/s", "ms_per_step": dt_f / max(args.steps // 2, 5) * 1e3,
                               "mfma_roofline_frac_step": v * FLOP_TRAIN_PER_SAMPLE / (PEAK_BF16_MFMA_TFLOPS * 1e12),
                               "note": "same step with plain bf16 MFMA operands; NOT a 1e-4 parity mode "
                                       "(observed errors: tests/test_hip_render.py::test_fast_mode_reports_error)"}

    if rank == 0 and world == 1 and not args.no_fast_extra and standard:
        # the womask configuration (cfg 5 shape): + 32 background samples per ray through the
        # NeRF++ kernels (K7); cos_anneal_ratio ramps there (a device scalar of the replayed step)
        try:
            import copy
            from fneus.trainer import WMASK_MODEL
            conf = copy.deepcopy(WMASK_MODEL)
            conf["neus_renderer"]["n_outside"] = 32
            trw = Stage1Trainer(device, model_conf=conf, prec=prec, use_graph=not args.no_graph)
            wb = synthetic_batches(14, RAYS, device, rank=rank)
            bg = torch.ones(1, 3, device=device)
            for i, b in enumerate(wb[:4]):
                trw.train_step(b, cos_anneal_ratio=0.01 * i, background_rgb=bg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i, b in enumerate(wb[4:]):
                trw.train_step(b, cos_anneal_ratio=0.04 + 0.01 * i, background_rgb=bg)
            torch.cuda.synchronize()
            dt_w = (time.perf_counter() - t0) / 10
            result["womask_step"] = {"value": RAYS * (N_SAMPLES + N_IMPORTANCE + 32) / dt_w, "unit": "ray-samples/s",
                                     "ms_per_step": dt_w * 1e3,
                                     "note": "512 rays x (64+64 inside + 32 outside) samples, background NeRF++ on the fused K7 "
                                             "kernels, ramping cos_anneal_ratio, same precision mode and launch mode as the headline number"}
            # BASELINE configs[4] quotes 2048 rays per batch: the same step on a 2048-ray batch (a new graph for the new shape)
            try:
                trw2 = Stage1Trainer(device, model_conf=conf, prec=prec, use_graph=not args.no_graph)
                wb2 = synthetic_batches(12, 2048, device, rank=rank, seed0=5000)
                for i, b in enumerate(wb2[:4]):
                    trw2.train_step(b, cos_anneal_ratio=0.01 * i, background_rgb=bg)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i, b in enumerate(wb2[4:]):
                    trw2.train_step(b, cos_anneal_ratio=0.04 + 0.01 * i, background_rgb=bg)
                torch.cuda.synchronize()
                dt_w2 = (time.perf_counter() - t0) / 8
                result["womask_step"]["rays_2048"] = {"value": 2048 * (N_SAMPLES + N_IMPORTANCE + 32) / dt_w2, "unit": "ray-samples/s",
                                                      "ms_per_step": dt_w2 * 1e3}
                del trw2
            except Exception as e:   # an extra must never take the headline number down with it
                result["womask_step"]["rays_2048"] = {"value": None, "error": repr(e)}
            del trw
        except Exception as e:   # an extra must never take the headline number down with it
            result["womask_step"] = {"value": None, "error": repr(e)}

    if rank == 0 and world == 1 and not args.no_fast_extra and standard:
        # stage 2 (BASELINE configs[2], lvis.py:132-196): 512 primary rays x (64+64), 4 secondary rays per hit point x 512
        # coarse SDF samples on K1, Lvis + IndirectLight trained with Adam.  Fixed-shape step (every ray treated as a hit point,
        # masked afterwards) replayed as one hipGraph, like the headline step.
        # stage 3 (BASELINE configs[3], mateIllu.py:135-203): 512 primary rays, 128 light SGs x 32 directions = 4096 Lvis
        # evaluations per hit point, closed-form SG rendering, Adam over the EnvmapMaterialNetwork
        sb = synthetic_batches(14, RAYS, device, rank=rank)

        def timed_stage(make_trainer):
            """10 timed steps after 4 warm-ups; the hit count is averaged over the timed steps (graph mode returns the same
            static tensor every replay, so it is accumulated on the device step by step)"""
            tr = make_trainer()
            for b in sb[:4]:
                tr.train_step(b)
            torch.cuda.synchronize()
            hit_sum = torch.zeros((), device=device, dtype=torch.float64)
            n_counted = 0
            t0 = time.perf_counter()
            for b in sb[4:]:
                o = tr.train_step(b)
                if o is not None:
                    hit_sum += o["n_hit"].to(torch.float64).reshape(())
                    n_counted += 1
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            return dt, (float(hit_sum) / n_counted if n_counted else 0.0), bool(tr.use_graph)

        STAGE_PMC_FILES = {"stage2": ("r06_t_stage2_pmc.json", "r06_u_stage2_pmc.json", "r06_v_stage2_pmc.json", "r06_w_stage2_pmc.json", "r06_z_stage2_pmc.json", "r06_y_stage2_pmc.json", "r06_x_stage2_pmc.json"),
                           "stage3": ("r06_t_stage3_pmc.json", "r06_u_stage3_pmc.json", "r06_v_stage3_pmc.json", "r06_w_stage3_pmc.json", "r06_z_stage3_pmc.json", "r06_y_stage3_pmc.json", "r06_x_stage3_pmc.json")}
        STAGE_KERNEL = {"fneus_sdf_fwd_rays": "sdf_fwd_p2_kernel", "fneus_sdf_fwd": "sdf_fwd_p2_kernel", "fneus_lvis_visibility": "lvis_visibility_p2_kernel"}

        def stage_roofline(stage, make_trainer):
            """the stage's dominant kernel: its launch duration measured HERE (HIP events around the eager fixed-shape step's launches),
            its issued MFMA FLOP and HBM bytes per launch from the committed PMC passes (tools/collect_stage_pmc.sh)"""
            tr = make_trainer()
            for b in sb[:2]:
                tr._fixed_shape_step(b)
            ops.profile_begin()
            for b in sb[2:4]:
                tr._fixed_shape_step(b)
            pr = ops.profile_end()
            name = max((k for k in pr if k in STAGE_KERNEL), key=lambda k: pr[k][1])
            n_l, ms = pr[name]
            avg_s = ms / n_l * 1e-3
            e = {"kernel": name, "launches_per_step": n_l / 2.0, "avg_launch_ms": ms / n_l, "ms_per_step": ms / 2.0}
            for tag in STAGE_PMC_FILES[stage]:
                try:
                    pj = json.load(open(os.path.join(ROOT, "profiles", tag)))
                    kname = next(k for k in pj["kernels"] if k.startswith(STAGE_KERNEL[name]))
                    pk = pj["kernels"][kname]
                    issued = pk["mfma_flop_issued_per_launch"]
                    e.update({"bound": "mfma", "achieved": issued / 3.0 / avg_s / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                              "frac": issued / 3.0 / avg_s / 1e12 / PEAK_BF16_MFMA_TFLOPS, "mfma_issue_frac": issued / avg_s / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                              "traffic": pk.get("hbm_bytes_per_launch"), "mfma_busy_pmc": pk.get("mfma_busy"),
                              "traffic_source": "profiles/" + tag + " (" + kname + ")",
                              "note": "algorithmic FLOP = issued bf16 MFMA FLOP (SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512, committed PMC pass) / 3 products per "
                                      "value, over this run's HIP-event launch duration; mfma_busy_pmc = the matrix pipe's busy fraction in that pass"})
                    break
                except Exception:
                    continue
            return e

        try:
            from fneus.trainer2 import Stage2Trainer
            dt_2, n_hit, graph2 = timed_stage(lambda: Stage2Trainer(device, prec=prec, use_graph=not args.no_graph))
            result["stage2_step"] = {"value": 4 * n_hit * 512 / dt_2, "unit": "secondary-ray SDF samples/s", "ms_per_step": dt_2 * 1e3,
                                     "primary_rays": RAYS, "mean_hit_points": n_hit, "secondary_rays": 4 * n_hit,
                                     "launch": "one hipGraph replay per step (fixed shape: the 4 secondary rays of every primary ray with a hit marched)"
                                               if graph2 else "eager launches, hit points compacted",
                                     "note": "lvis_render + L1 losses + backward + Adam (lvis.py:132-196): 4 secondary rays per hit "
                                             "point x 512 coarse samples through K1, 32 fine samples through K2, same precision mode"}
            try:
                result["stage2_step"]["roofline"] = stage_roofline("stage2", lambda: Stage2Trainer(device, prec=prec, use_graph=False))
            except Exception as e:
                result["stage2_step"]["roofline"] = {"error": repr(e)}
        except Exception as e:   # an extra must never take the headline number down with it
            result["stage2_step"] = {"value": None, "error": repr(e)}
        try:
            from fneus.trainer3 import Stage3Trainer
            dt_3, n_hit3, graph3 = timed_stage(lambda: Stage3Trainer(device, prec=prec, use_graph=not args.no_graph))
            result["stage3_step"] = {"value": 4096 * n_hit3 / dt_3, "unit": "Lvis visibility evaluations/s", "ms_per_step": dt_3 * 1e3,
                                     "primary_rays": RAYS, "mean_hit_points": n_hit3,
                                     "launch": "one hipGraph replay per step (fixed shape: all 512 rays evaluated, misses masked)"
                                               if graph3 else "eager launches, hit points compacted",
                                     "note": "mateIllu_render + masked L1 + latent sparsity + backward + Adam (mateIllu.py:135-203): "
                                             "128 light lobes x 32 directions per hit point (the nominal count `value` uses; the network "
                                             "runs on the directions that face the surface, as in the reference: about half), SG rendering "
                                             "of 128 + 24 lobes"}
            try:
                result["stage3_step"]["roofline"] = stage_roofline("stage3", lambda: Stage3Trainer(device, prec=prec, use_graph=False))
            except Exception as e:
                result["stage3_step"]["roofline"] = {"error": repr(e)}
        except Exception as e:
            result["stage3_step"] = {"value": None, "error": repr(e)}
        try:    # the lever this step has left, measured and NOT the default: its visibility launch with ONE fp16 product
            def _s3_h16():
                tr3 = Stage3Trainer(device, prec=prec, use_graph=not args.no_graph)
                tr3.lvis_network.set_precision(ops.PREC_H16)
                return tr3
            dt_3h, _, _ = timed_stage(_s3_h16)
            result["stage3_step_lvis_one_fp16_product"] = {
                "ms_per_step": dt_3h * 1e3,
                "note": "Lvis.set_precision(ops.PREC_H16) / FNEUS_LVIS_PREC=2: the visibility kernel at a third of its matrix work (1.25 -> 0.57 ms). "
                        "On the synthetic network every output of the reference fixtures stays within 3.5e-5 (a lobe's visibility within 3.9e-5; "
                        "three bf16 products: 3e-7), but the error grows with the network's sharpness (fp64 emulation, "
                        "tools/experiments/r05/lvis_schemes.py: 1.5e-4 with every hidden layer's weights doubled) -- not a parity mode, off by default"}
        except Exception as e:
            result["stage3_step_lvis_one_fp16_product"] = {"ms_per_step": None, "error": repr(e)}

    if rank == 0 and world == 1 and not args.no_fast_extra and standard and prec == ops.PREC_PARITY:
        try:    # the same step with fp32-accurate weight gradients (hi + lo planes): the other pinned mode (tests/test_hip_render.py)
            dt_g, _, _ = run(prec, max(args.steps // 2, 5), 3, profile=False, gprec=3)
            n_g = max(args.steps // 2, 5)
            result["exact_gradients_gprec3"] = {"value": SAMPLES_PER_STEP * n_g / dt_g, "unit": "ray-samples/s", "ms_per_step": dt_g / n_g * 1e3}
        except Exception as e:
            result["exact_gradients_gprec3"] = {"value": None, "error": repr(e)}
        try:    # gradient precision 1 (the default of rounds 2-5): bf16 planes for every product, the colour network's output layer included --
            # its gradient bounds are looser (tests/test_hip_render.py: 8e-3 / 5e-3 of scale; the 12-step loss trajectory 25 %)
            dt_m, _, _ = run(prec, max(args.steps // 2, 5), 3, profile=False, gprec=1)
            n_m = max(args.steps // 2, 5)
            result["bf16_gradient_planes_gprec1"] = {"value": SAMPLES_PER_STEP * n_m / dt_m, "unit": "ray-samples/s", "ms_per_step": dt_m / n_m * 1e3,
                                                     "note": "rounds 2-5 reported this mode as the headline; the headline's mode (gradient precision 2) "
                                                             "holds the gradient bounds of gprec 3 (5e-3 of scale per sampled element, 2e-3 of a tensor's norm)"}
        except Exception as e:
            result["bf16_gradient_planes_gprec1"] = {"value": None, "error": repr(e)}
        try:    # the headline's mode with hi + lo cotangents inside the backward chains (three MFMAs per product there, as before round 6)
            os.environ["FNEUS_BWD_XHI"] = "0"
            os.environ["FNEUS_COLB_XHI"] = "0"
            n_x = max(args.steps // 2, 5)
            dt_x, _, _ = run(prec, n_x, 3, profile=False)
            result["hi_lo_cotangent_chains"] = {"value": SAMPLES_PER_STEP * n_x / dt_x, "unit": "ray-samples/s", "ms_per_step": dt_x / n_x * 1e3,
                                                "note": "FNEUS_BWD_XHI=0 FNEUS_COLB_XHI=0: K3 and the colour backward with hi + lo activation fragments "
                                                        "(the gradients of the 512-ray fixture: 2.8e-4 / 2.7e-5 of scale / norm against 5.7e-4 / 3.2e-5 "
                                                        "with the default's bf16 cotangents, bounds 5e-3 / 2e-3; DESIGN.md 4.1e)"}
        except Exception as e:
            result["hi_lo_cotangent_chains"] = {"value": None, "error": repr(e)}
        finally:
            os.environ.pop("FNEUS_BWD_XHI", None)
            os.environ.pop("FNEUS_COLB_XHI", None)
        try:    # config 5's per-GPU share of a 2048-ray batch at 8 ranks: 256 rays (womask shape), the strong-scaling point
            import copy
            from fneus.trainer import WMASK_MODEL
            conf = copy.deepcopy(WMASK_MODEL)
            conf["neus_renderer"]["n_outside"] = 32
            trs = Stage1Trainer(device, model_conf=conf, prec=prec, use_graph=not args.no_graph)
            sb_ = synthetic_batches(16, 256, device, rank=rank, seed0=7000)
            bg_ = torch.ones(1, 3, device=device)
            for b in sb_[:5]:
                trs.train_step(b, cos_anneal_ratio=0.5, background_rgb=bg_)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in sb_[5:]:
                trs.train_step(b, cos_anneal_ratio=0.5, background_rgb=bg_)
            torch.cuda.synchronize()
            dt_s = (time.perf_counter() - t0) / 11
            result["womask_256_rays_step"] = {"value": 256 * 160 / dt_s, "unit": "ray-samples/s", "ms_per_step": dt_s * 1e3,
                                              "note": "256 rays x (64+64+32): one rank's share of configs[4]'s 2048-ray batch at 8 ranks "
                                                      "(python bench.py --gpus 8 --rays-global 2048 --womask measures the 8-rank job)"}
            del trs
        except Exception as e:
            result["womask_256_rays_step"] = {"value": None, "error": repr(e)}

    if rank == 0 and world == 1 and standard and prec == ops.PREC_PARITY and not args.no_fast_extra:
        # observed parity errors of THIS build on THIS box against the reference's own outputs: the
        # 512-ray x (64+64) fixture tests/golden/render_wmask_b512_n64.npz through the helpers of tests/test_hip_render.py
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import test_hip_render as thr
            g = thr.load(os.path.join(ROOT, "tests", "golden"), "render_wmask_b512_n64")
            out_t, _, _ = thr.run(g, 3, teacher_z=True)
            par = {k: thr.maxerr(thr.stored(out_t[k], g) if (out_t[k].dim() >= 2 and out_t[k].shape[1] not in (1, 3)) else out_t[k],
                                 g["out/" + k]) for k in ("color_fine", "weights", "gradients", "weight_sum", "surface_color")}
            par["sdf"] = thr.maxerr(thr.stored(out_t["_sdf"], g), g["core/sdf"])
            out_e, _, _ = thr.run(g, 3, teacher_z=False)
            dz = (out_e["_z_vals"].detach().cpu() - thr.final_z(g)).abs()
            result["parity"] = {"fixture": "tests/golden/render_wmask_b512_n64.npz (the reference's own render of 512 rays x (64+64))",
                                "max_abs_error_teacher_forced_z": par,
                                "z_vals_own_sampler": {"max_abs": float(dz.max()), "frac_within_1e-4": float((dz <= 1e-4).float().mean())},
                                "color_fine_own_sampler_max_abs": thr.maxerr(out_e["color_fine"], g["out/color_fine"]),
                                "tolerance": 1e-4}
        except Exception as e:
            result["parity"] = {"error": repr(e)}
        try:
            cj = json.load(open(os.path.join(ROOT, "profiles", "r06_chamfer.json" if os.path.exists(os.path.join(ROOT, "profiles", "r06_chamfer.json")) else "r05_chamfer.json")))
            result["chamfer"] = {k: cj[k] for k in ("what", "steps", "seeds", "hip_mean", "hip_sd", "oracle_mean", "oracle_sd",
                                                    "ratio_of_means", "sem_log_ratio_pct", "within_2_pct") if k in cj}
            result["chamfer"]["source"] = ("profiles/r06_chamfer.json (tests/checkers/chamfer_study.py on the builder's box, round-6 kernels at the default gradient "
                                           "precision 2; not measured in this run)" if os.path.exists(os.path.join(ROOT, "profiles", "r06_chamfer.json")) else
                                           "profiles/r05_chamfer.json (tests/checkers/chamfer_study.py on the builder's box, round-5 kernels, gradient precision 1; "
                                           "not measured in this run)")
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline()
            try:
                result["cpu_baseline"]["same_port_on_this_gpu"] = cpu_baseline(device=device)
            except Exception as e:
                result["cpu_baseline"]["same_port_on_this_gpu"] = {"value": None, "error": repr(e)}
        except Exception as e:   # the baseline must never take the GPU number down with it
            result["cpu_baseline"] = {"value": None, "error": repr(e)}

    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
