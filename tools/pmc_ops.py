"""Turn the per-shape rocprofv3 --pmc passes of tools/pmc_ops.sh into profiles/<round>_traffic.json.

    python tools/pmc_ops.py <pmc_dir> <round>      (e.g. gpurun_out/r2/pmc_b r02)

Per shape: kernel duration (from the counter rows' timestamps), MFMA-busy share, HBM-side bytes.
  * mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz): the share of the launch during which a
    SIMD's matrix pipe is busy, priced at the 2.4 GHz peak clock (so it is directly comparable with flops / 2.5 PFLOP/s).
    The counter adds, over all waves, the pipe cycles of every MFMA (32 per 32x32x16, 16 per 16x16x32); the calibration
    kernel (tools/micro/mfma_peak.hip: back-to-back MFMAs on every SIMD) reads 100 % of its MFMA cycles and fixes the unit.
  * mfma_busy_at_clock uses GRBM_GUI_ACTIVE / 8 (cycles the chip actually clocked, per XCD) instead of duration x 2.4 GHz;
    only meaningful for launches of >= 0.3 ms (the counter reads high on short dispatches), given for the long ones.
  * hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md).
"""
import csv
import glob
import json
import os
import sys

ES2 = 2.0 if os.environ.get("ST_BENCH_DTYPE", "bf16") == "fp32" else 1.0      # algorithmic bytes below are written for 2-byte elements: x2 for the strict mode's fp32 tensors
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "pmc_ops")
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
KEEP = ("attn", "gemm", "conv_halo", "conv_thin", "gn_", "mfma_peak")      # (the xattn launches are gemm_dma_kernel<..., XA>)

# shape -> (family, algorithmic flops, algorithmic bytes, launches per denoise step at bs=1 / latent 128)
def meta(name):
    p = name.split("_")
    if name.startswith("attn"):
        b = 4 if name.endswith("b4") else 1
        T, S, H = {"attn_self_4096": (4096, 4096, 10), "attn_self_1024": (1024, 1024, 20), "attn_cross_4096": (4096, 77, 10),
                   "attn_cross_1024": (1024, 77, 20), "attn_self_1024_b4": (1024, 1024, 20), "attn_self_4096_b4": (4096, 4096, 10)}[name]
        fam = "attention_self" if "self" in name else "attention_cross"
        cnt = {"attn_self_4096": 10, "attn_self_1024": 60, "attn_cross_4096": 10, "attn_cross_1024": 60}.get(name, 0)
        return fam, 4.0 * b * H * T * S * 64, ES2 * 2.0 * b * H * 64 * (2 * T + 2 * S), cnt
    if name.startswith("xattn"):
        M, C = (int(v) for v in p[1].split("x"))
        H = C // 64
        cnt = {"xattn_1024x1280": 60, "xattn_4096x640": 10}.get(name, 0)
        return "linear_xattn", 2.0 * M * C * C + 4.0 * M * 77 * C, 2.0 * (M * C + C * C + M * C + 2 * 77 * C), cnt
    if name.startswith("linear"):
        M, K, N = (int(v) for v in p[1].split("x"))
        g = 2 if "g" in (p[2] if len(p) > 2 else "") else 1
        cnt = {"linear_1024x1280x5120_lng": 60, "linear_1024x5120x1280": 60, "linear_1024x1280x1280": 192, "linear_1024x1280x3840_ln": 60,
               "linear_4096x640x640": 40, "linear_4096x640x2560_lng": 10, "linear_4096x2560x640": 10}.get(name, 0)
        return "linear", 2.0 * M * K * N * g, ES2 * 2.0 * (M * K + g * N * K + M * N), cnt
    if name.startswith("conv"):
        C, H = int(p[1]), int(p[2])
        cnt = {"conv_1280_32": 10, "conv_640_64": 6, "conv_320_128": 7}.get(name, 0)
        return "conv2d", 2.0 * H * H * C * C * 9, ES2 * 2.0 * (2 * H * H * C + 9 * C * C), cnt
    if name.startswith("gn"):
        C, H = int(p[1]), int(p[2])
        cnt = {"gn_320_128": 8, "gn_1280_32": 16, "gn_640_64": 11}.get(name, 0)
        return "group_norm", 0.0, ES2 * 2.0 * 2 * H * H * C, cnt
    return "calib", 0.0, 0.0, 0


def read(path_glob):
    out = {}
    for f in glob.glob(path_glob):
        for r in csv.DictReader(open(f)):
            if not any(k in r["Kernel_Name"] for k in KEEP):
                continue
            d = out.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"][:90], "ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(out.values())


shapes = sorted({os.path.basename(d).rsplit("__", 1)[0] for d in glob.glob(os.path.join(src, "*__*"))})
per, fam_acc = {}, {}
for name in shapes:
    fam, flops, nbytes, cnt = meta(name)
    # (one_op.py runs the producer GEMM once before the LayerNorm-folded launches: keep only the kernel under study)
    # r02 kept that producer launch in the LayerNorm-folded linear shapes (1 of 6 dispatches, a short 1280^2 GEMM): it pulled the
    # average duration down and with it mfma_busy below flops_over_peak by 12-15 %.  Keep the kernel that was launched most.
    def only(ds):
        if fam == "linear_xattn":
            return [d for d in ds if "ELb1ELb1EEv" in d["kernel"]]
        if fam in ("linear", "conv2d") and ds:
            names = {}
            for d in ds:
                names[d["kernel"]] = names.get(d["kernel"], 0) + 1
            top = max(names.values())
            # (round 5: a call may be TWO kernels - the column split of dispatch.h: full rounds on the eight-phase kernel + the remaining
            #  columns on smaller tiles - so every kernel launched as often as the most-launched one belongs to the call)
            return [d for d in ds if names[d["kernel"]] == top]
        return ds
    sq = only(read(os.path.join(src, name + "__sq", "*", "*counter_collection.csv")))
    if not sq:
        continue
    # one launch of an operator may be several kernels (GroupNorm: three): launches = dispatches / kernels per call
    kernels = sorted({d["kernel"] for d in sq})
    calls = max(1, len(sq) // max(1, len(kernels))) if fam in ("group_norm", "linear", "conv2d") else len(sq)
    # round 5: drop a cold first dispatch (code pages of a kernel never run in the process: one launch of 127 us among four of
    # 22.9 us doubled the "average" of a shape) - a single-kernel shape with >= 4 dispatches loses every dispatch that took
    # more than twice the median
    if fam != "group_norm" and len(kernels) == 1 and len(sq) >= 4:
        med = sorted(d["ns"] for d in sq)[len(sq) // 2]
        kept = [d for d in sq if d["ns"] <= 2.0 * med]
        if len(kept) >= 3:
            sq, calls = kept, len(kept)
    ns = sum(d["ns"] for d in sq) / calls
    mfma = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in sq) / calls
    grbm = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in sq) / calls / 8.0
    rec = {"family": fam, "kernels": kernels, "us_per_launch_under_pmc": round(ns / 1e3, 2), "launches_per_step": cnt,
           "mfma_busy": round(mfma / (1024.0 * ns * 2.4), 4) if ns else None}
    if ns >= 3e5:
        rec["mfma_busy_at_clock"] = round(mfma / (1024.0 * grbm), 4)
        rec["clock_ghz"] = round(grbm / ns, 3)
    if flops:
        rec["algorithmic_flops"] = flops
        rec["flops_over_peak"] = round(flops / (ns * 1e-9) / 2.5e15, 4)
    fetch = only(read(os.path.join(src, name + "__fetch", "*", "*counter_collection.csv")))
    write = only(read(os.path.join(src, name + "__write", "*", "*counter_collection.csv")))
    if fetch and write:
        # (per call of THEIR passes: the duration filter above may have dropped a cold dispatch from the SQ pass only)
        multi = fam in ("group_norm", "linear", "conv2d")
        fcalls = max(1, len(fetch) // max(1, len(kernels))) if multi else len(fetch)
        wcalls = max(1, len(write) // max(1, len(kernels))) if multi else len(write)
        fs = sum(d.get("FETCH_SIZE", 0.0) for d in fetch) / fcalls
        wsz = sum(d.get("WRITE_SIZE", 0.0) for d in write) / wcalls
        rec["FETCH_SIZE"], rec["WRITE_SIZE"] = round(fs, 1), round(wsz, 1)
        rec["hbm_side_bytes"] = int((2 * fs + wsz) * 1024)
        rec["algorithmic_bytes"] = int(nbytes)
        rec["traffic_ratio"] = round(rec["hbm_side_bytes"] / nbytes, 2) if nbytes else None
    per[name] = rec
    if cnt and fam != "calib":
        a = fam_acc.setdefault(fam, {"n": 0, "bytes": 0.0, "alg": 0.0, "mfma_ns": 0.0, "ns": 0.0, "flops": 0.0})
        a["flops"] += flops * cnt
        a["n"] += cnt
        a["bytes"] += rec.get("hbm_side_bytes", 0) * cnt
        a["alg"] += nbytes * cnt
        a["mfma_ns"] += rec["mfma_busy"] * ns * cnt
        a["ns"] += ns * cnt

out = {"_what": "hardware counters per operator shape, MI355X, rocprofv3 --pmc (tools/pmc_ops.sh, one shape per process, 5 launches averaged)",
       "_how": __doc__.split("Per shape:")[1].strip(), "per_shape": per}
for fam, a in fam_acc.items():
    out[f"{fam}_bytes_per_launch"] = int(a["bytes"] / a["n"])
    out[f"{fam}_algorithmic_bytes_per_launch"] = int(a["alg"] / a["n"])
    out[f"{fam}_mfma_busy"] = round(a["mfma_ns"] / a["ns"], 4)
    if a["flops"]:      # USEFUL matrix work over the dense bf16 peak (mfma_busy also counts the attention's ones-block row sums: 4 of its 20 MFMAs per tile)
        out[f"{fam}_flops_over_peak"] = round(a["flops"] / (a["ns"] * 1e-9) / 2.5e15, 4)
path = os.path.join(root, "profiles", f"{rnd}_traffic.json")
json.dump(out, open(path, "w"), indent=1)
for k, v in per.items():
    print(f"{k:34s} {v['us_per_launch_under_pmc']:8.1f} us  mfma_busy {v['mfma_busy']}  at_clock {v.get('mfma_busy_at_clock')}  "
          f"traffic {v.get('hbm_side_bytes')} / {v.get('algorithmic_bytes')} = {v.get('traffic_ratio')}")
print(path)
