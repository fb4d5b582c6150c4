#!/usr/bin/env python3
"""rocprofv3 --pmc passes (directories) -> JSON: per fneus kernel the average counter values per launch and what follows from them:
hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE under-counts wide coalesced reads 2x on gfx950, MI355X_MICROARCH.md), issued
MFMA FLOP = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 (calibrated on K3: 64 per v_mfma_f32_32x32x16_bf16), MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES
/ (SQ_BUSY_CYCLES / 32 x 1024).   python tools/pmc_kernels_json.py --steps N --out file.json dir [dir ...]"""
import collections, csv, glob, json, sys
args = sys.argv[1:]
steps = int(args[args.index("--steps") + 1]); out = args[args.index("--out") + 1]
dirs = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] not in ("--steps", "--out"))]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in dirs:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "fneus::" not in name:
                continue
            acc[name.split("fneus::")[1].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    e = {"launches_per_step": max(len(v) for v in cs.values()) / steps, "counters_per_launch": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in c:
        e["mfma_flop_issued_per_launch"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
    if c.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
        e["wait_inst_per_wave_cycle"] = c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    res[k] = e
order = sorted(res, key=lambda k: -res[k].get("mfma_flop_issued_per_launch", 0) * res[k]["launches_per_step"])
json.dump({"_note": __doc__, "steps": steps, "kernels": {k: res[k] for k in order}}, open(out, "w"), indent=1)
for k in order[:8]:
    e = res[k]
    print(f"{k[:60]:60s} x{e['launches_per_step']:5.1f}  hbm {e.get('hbm_bytes_per_launch', 0) / 1e6:9.1f} MB  issued {e.get('mfma_flop_issued_per_launch', 0) / 1e9:9.1f} GFLOP  "
          f"busy {100 * e.get('mfma_busy', 0):5.1f} %")
