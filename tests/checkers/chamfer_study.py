#!/usr/bin/env python3
"""Chamfer-L1 at equal steps, HIP path vs the oracle, over several seeds (BASELINE.json metric, second half).

Scene: the analytic two-sphere scene of models/dataset.py (DTU-shaped cameras on a sphere, masks); per seed BOTH paths start
from the same weights (fneus.synth streams of that seed), see the same ray batches and the same learning-rate schedule
(warm-up + cosine, exp_runner.py:229-238) and train for the same number of steps; then the zero level set of each SDF is
extracted on a grid (K1 for the HIP path, the oracle's own sdf for the oracle) and scored against the analytic surface with
evaluation/chamfer.py (the restatement of evaluation/dtu_eval.py:36-162, pinned to the reference by tests/test_mesh_cpu.py).
The HIP path runs in deterministic mode (fneus_dw_gemm_pp_det), so a seed gives one number, not a distribution.

Usage: chamfer_study.py [--seeds 8] [--steps 2000] [--rays 512] [--res 128] [--out profiles/r03_chamfer.json]
The oracle runs on the same GPU through stock PyTorch-ROCm ops (27 ms per step at 512 x 128)."""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "factored-neus_amd"))
import numpy as np
import torch


def lr_at(it, steps, lr0=5e-4, alpha=0.05):
    wu = 0.1 * steps
    if it < wu:
        return lr0 * it / wu
    return lr0 * ((0.5 * (1 + np.cos(np.pi * (it - wu) / (steps - wu)))) * (1 - alpha) + alpha)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--rays", type=int, default=512)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_chamfer.json"))
    ap.add_argument("--gprec", type=int, default=2)
    ap.add_argument("--seed0", type=int, default=100)
    args = ap.parse_args()
    summary = run_study(args)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "runs"}))


def run_study(args):
    """args: seeds, steps, rays, res, gprec, seed0 (an argparse namespace or anything with these attributes)"""

    from evaluation.chamfer import evaluate_mesh
    from fneus import ops, synth
    from fneus.trainer import Stage1Trainer, WMASK_MODEL
    from models.dataset import SyntheticDataset, scene_surface_points
    from models.mesh import extract_fields, marching_tetrahedra
    from oracle import ref_torch as R

    dev = torch.device("cuda:0")
    ops.set_deterministic(True)
    gt = scene_surface_points(60000, seed=0)
    res = args.res

    def mesh(u):
        v, f = marching_tetrahedra(u, 0.0)
        return v.cpu().numpy().astype(np.float64) / (res - 1.0) * 2.02 - 1.01, f.cpu().numpy()

    def chamfer(u):
        v, f = mesh(u)
        return float(evaluate_mesh(v, f, gt, thresh=0.01, max_dist=1.0)[2]) if len(f) else float("nan")

    conf = copy.deepcopy(WMASK_MODEL)
    conf["neus_renderer"]["perturb"] = 0.0            # same depths on both sides (the jitter streams differ)
    rows = []
    t_start = time.time()
    for k in range(args.seeds):
        seed = getattr(args, "seed0", 100) + 7 * k
        ds = SyntheticDataset(n_images=24, H=192, W=256, device=dev, seed=1)
        rs = np.random.RandomState(seed)
        torch.manual_seed(seed)
        batches = [ds.gen_random_rays_at(int(rs.randint(ds.n_images)), args.rays) for _ in range(args.steps)]
        # ---- HIP path
        t0 = time.time()
        tr = Stage1Trainer(dev, model_conf=conf, prec=ops.PREC_PARITY, seed=seed, lr=5e-4, use_graph=True, gprec=args.gprec)
        c_init = chamfer(tr.renderer.extract_sdf_grid([-1.01] * 3, [1.01] * 3, res))
        for it, b in enumerate(batches, 1):
            tr.set_lr(lr_at(it, args.steps))
            tr.train_step(b)
        torch.cuda.synchronize()
        t_hip = time.time() - t0
        c_hip = chamfer(tr.renderer.extract_sdf_grid([-1.01] * 3, [1.01] * 3, res))
        del tr
        # ---- oracle: same weights, same batches, same schedule, torch.optim.Adam
        t0 = time.time()
        T = lambda sd: {n: torch.from_numpy(v).clone().to(dev).requires_grad_(True) for n, v in sd.items()}
        sd_sdf, sd_col, sd_ref = T(synth.sdf_state_dict(seed)), T(synth.color_state_dict(seed + 1)), T(synth.refcolor_state_dict(seed + 2))
        variance = torch.tensor(0.3, device=dev, requires_grad=True)
        opt = torch.optim.Adam(list(sd_sdf.values()) + list(sd_col.values()) + list(sd_ref.values()) + [variance], lr=5e-4)
        for it, b in enumerate(batches, 1):
            for gq in opt.param_groups:
                gq["lr"] = lr_at(it, args.steps)
            near, far = R.near_far_from_sphere(b[:, :3], b[:, 3:6])
            out = R.render(b[:, :3], b[:, 3:6], near, far, R.sdf_params_from_state_dict(sd_sdf), R.inv_s_from_variance(variance),
                           R.color_params_from_state_dict(sd_col), sd_ref, None, n_samples=64, n_importance=64, t_rand=None,
                           cos_anneal_ratio=1.0)
            losses = R.stage1_loss(out, b[:, 6:9], b[:, 9:10], 0.1, 0.1, 0.1)
            opt.zero_grad()
            losses["loss"].backward()
            opt.step()
        torch.cuda.synchronize()
        t_ref = time.time() - t0
        with torch.no_grad():
            p_sdf = R.sdf_params_from_state_dict(sd_sdf)
            u_ref = extract_fields([-1.01] * 3, [1.01] * 3, res, lambda pts: -R.sdf_only(pts, p_sdf).reshape(-1), device=dev, as_numpy=False)
        c_ref = chamfer(u_ref)
        rows.append({"seed": seed, "chamfer_init": c_init, "chamfer_hip": c_hip, "chamfer_oracle": c_ref, "train_s_hip": t_hip,
                     "train_s_oracle": t_ref})
        print(f"seed {seed}: initial {c_init:.4f}  HIP {c_hip:.4f} ({t_hip:.1f} s)  oracle {c_ref:.4f} ({t_ref:.1f} s)", flush=True)
    h = np.array([r["chamfer_hip"] for r in rows])
    o = np.array([r["chamfer_oracle"] for r in rows])
    n = len(rows)
    sem = lambda x: float(x.std(ddof=1) / np.sqrt(len(x))) if len(x) > 1 else float("nan")
    ratio = float(h.mean() / o.mean())
    # resolution of the comparison: standard error of the mean of the per-seed log ratios
    lr_ = np.log(h / o)
    summary = {
        "what": "Chamfer-L1 to the analytic two-sphere scene after equal steps, HIP path (deterministic mode, parity arithmetic, "
                f"gradient precision {args.gprec}) vs the oracle on the same weights / batches / schedule",
        "steps": args.steps, "rays": args.rays, "samples": "64+64", "grid": res, "seeds": n,
        "hip_mean": float(h.mean()), "hip_sd": float(h.std(ddof=1)) if n > 1 else None,
        "oracle_mean": float(o.mean()), "oracle_sd": float(o.std(ddof=1)) if n > 1 else None,
        "ratio_of_means": ratio, "ratio_minus_1_pct": 100.0 * (ratio - 1.0),
        "mean_log_ratio_pct": 100.0 * float(lr_.mean()), "sem_log_ratio_pct": 100.0 * sem(lr_),
        "within_2_pct": bool(abs(ratio - 1.0) <= 0.02),
        "resolved_at_2_pct": bool(100.0 * sem(lr_) <= 1.0),
        "runs": rows, "wall_s": time.time() - t_start,
    }
    ops.set_deterministic(None)
    return summary


if __name__ == "__main__":
    main()
