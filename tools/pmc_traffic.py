"""Turn the per-shape rocprofv3 --pmc passes of tools/pmc_traffic.sh (gpurun_out/pmc/) into profiles/<round>_traffic.json."""
import csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
# launches per denoise step (bs=1, latent 128) from bench.py's census (ST_CENSUS_SHAPES=1)
COUNTS = {"1024x1280x1280": 192, "1024x5120x1280": 60, "1024x1280x3840": 60, "1024x1280x5120g": 60,
          "4096x640x640": 40, "4096x640x1920": 10, "4096x640x2560g": 10, "4096x2560x640": 10}
per = {}
for shape, cnt in COUNTS.items():
    rec = {"launches_per_step": cnt}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(root, "gpurun_out", "pmc", f"{shape}_{c}", "*", "*counter_collection.csv"))
        vals = []
        for f in files:
            for r in csv.DictReader(open(f)):
                if "gemm_dma_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        rec[c] = round(sum(vals) / len(vals), 1) if vals else None
        rec[c + "_launches"] = len(vals)
    M, K, N = (int(v) for v in shape.rstrip("g").split("x"))
    g = 2 if shape.endswith("g") else 1
    rec["algorithmic_bytes"] = 2 * (M * K + g * N * K + M * N)
    rec["hbm_side_bytes"] = int((2 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024)
    per[shape] = rec
tot = sum(r["launches_per_step"] for r in per.values())
avg = sum(r["hbm_side_bytes"] * r["launches_per_step"] for r in per.values()) / tot
alg = sum(r["algorithmic_bytes"] * r["launches_per_step"] for r in per.values()) / tot
out = {
    "_what": "HBM-side traffic of the Linear GEMM kernel family (gemm_dma_kernel<bf16,...,CONV=false>), MI355X",
    "_how": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE around "
            "tools/one_gemm.py <M K N [g]> (5 launches averaged); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE "
            "counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM). bench.py's full-model run segfaults under --pmc in this "
            "ROCm build, so the family average is the launch-count-weighted mean of the per-shape values (counts per denoise step, "
            "bs=1, latent 128).",
    "per_shape": per,
    "linear_bytes_per_launch": int(avg),
    "linear_algorithmic_bytes_per_launch": int(alg),
    "ratio": round(avg / alg, 2),
}
path = os.path.join(root, "profiles", f"{rnd}_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(path, "avg", int(avg), "alg", int(alg), "ratio", round(avg / alg, 2))
for k, r in per.items():
    print(k, r["hbm_side_bytes"], r["algorithmic_bytes"], round(r["hbm_side_bytes"] / r["algorithmic_bytes"], 2), r["FETCH_SIZE_launches"])
