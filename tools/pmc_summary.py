#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection.csv values per kernel (fneus kernels only)."""
import csv, sys, collections, glob, json
json_out, n_steps = None, None
if "--steps" in sys.argv:           # number of identical training steps the profiled run made: per-step totals
    i = sys.argv.index("--steps")
    n_steps = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    json_out = sys.argv[i + 1]
    del sys.argv[i:i + 2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "fneus::" not in name: continue
            short = name.split("fneus::")[1].split("(")[0]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")

# matrix-pipe utilisation where the SQ counters were collected: SQ_BUSY_CYCLES counts per shader engine (32 on this part),
# SQ_VALU_MFMA_BUSY_CYCLES per SIMD (256 CUs x 4): busy fraction = MFMA_BUSY / (BUSY / 32 x 1024)
rows = []
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CYCLES", 0) > 0:
        simd_cycles = c["SQ_BUSY_CYCLES"] / 32.0 * 1024.0
        rows.append((c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, k, c))
if rows:
    print("# matrix-pipe utilisation per kernel (average launch): MFMA busy cycles / SIMD cycles; waves waiting on an instruction")
    print("# %-44s %10s %14s %12s" % ("kernel", "MFMA busy", "wait_inst/wave", "waves"))
    for u, k, c in sorted(rows, reverse=True):
        wi = c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        print("# %-44s %9.1f%% %13.1f%% %12.0f" % (k[:44], 100 * u, 100 * wi, c.get("SQ_WAVES", 0)))

if json_out:
    # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE under-counts wide coalesced reads 2x on gfx950
    # (MI355X_MICROARCH.md, HBM section).  Keys are the C-ABI entry points bench.py reports its kernels under.
    entry = {"sdf_bwd": "fneus_sdf_bwd", "sdf_fwd_grad": "fneus_sdf_fwd_grad", "sdf_grad_rev": "fneus_sdf_fwd_grad",
             "dw_gemm_pp": "fneus_dw_gemm_pp",
             "color_fwd": "fneus_color_fwd", "color_bwd": "fneus_color_bwd", "sdf_fwd": "fneus_sdf_fwd",
             "refcolor_fwd": "fneus_refcolor_fwd", "refcolor_bwd": "fneus_refcolor_bwd"}
    ks = {}
    for k in sorted(acc):
        for pre, name in entry.items():
            if not k.startswith(pre):
                continue
            if name not in ks and "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
                f = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
                w = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
                ks[name] = {"kernel": k, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
            break           # first entry wins: sdf_fwd_grad_* must not also file under sdf_fwd
    # entry points that are several launches: K2 = forward chain with the stash + reverse sweep (round 3)
    for name, pre in (("fneus_sdf_fwd_grad", "sdf_fwd_stash_p2"),):
        for k in sorted(acc):
            if k.startswith(pre) and name in ks and "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k] and k not in ks[name]["kernel"]:
                f = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
                w = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
                e = ks[name]
                ks[name] = {"kernel": e["kernel"] + " + " + k, "FETCH_SIZE_KB": e["FETCH_SIZE_KB"] + f, "WRITE_SIZE_KB": e["WRITE_SIZE_KB"] + w,
                            "hbm_bytes_per_launch": e["hbm_bytes_per_launch"] + (2 * f + w) * 1024}
    step = None
    if n_steps:
        tot_f = sum(sum(acc[k]["FETCH_SIZE"]) for k in acc if "FETCH_SIZE" in acc[k])
        tot_w = sum(sum(acc[k]["WRITE_SIZE"]) for k in acc if "WRITE_SIZE" in acc[k])
        per_k = {k: (2 * sum(acc[k].get("FETCH_SIZE", [0])) + sum(acc[k].get("WRITE_SIZE", [0]))) * 1024 / n_steps for k in acc}
        step = {"steps": n_steps, "hbm_bytes_per_step": (2 * tot_f + tot_w) * 1024 / n_steps,
                "bytes_per_ray_sample": (2 * tot_f + tot_w) * 1024 / n_steps / 65536,
                "by_kernel_bytes_per_step": dict(sorted(per_k.items(), key=lambda kv: -kv[1]))}
        print(f"# whole step: {step['hbm_bytes_per_step'] / 1e9:.3f} GB = {step['bytes_per_ray_sample'] / 1024:.1f} KiB per ray sample")
    json.dump({"step_total": step, "_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh, "
                        "tools/pmc_run.py parity 2, N = 65536 samples per launch); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
                        "FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section). "
                        "A kernel with several launch shapes per step (dw_gemm, sdf_fwd) is the average over them.",
               "kernels": ks}, open(json_out, "w"), indent=1)
