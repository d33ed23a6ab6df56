"""Per-family kernel time per denoise step from a rocprofv3 `--kernel-trace --stats` summary of bench.py.

    python tools/profile_families.py profiles/r02_bench_kernel_stats.csv [out.json]

Families are the ones bench.py's roofline names.  The number of UNet steps in the run is the call count of
conv_thin_kernel (conv_in: exactly one launch per step); one-off launches (text-context / time-table GEMMs per prompt,
capture warm-up) are part of the run and stay in the totals - with >= 100 replayed steps they are < 1 %.
"""
import csv
import json
import os
import re
import sys


def family(name: str):
    if "conv_halo_kernel" in name or "conv_thin_kernel" in name:
        return "conv2d"
    # gemm_dma_kernel<T, BM, BN, WGM, WGN, STAGES, U, CONV, GEGLU, LNF, XA>: mangled (what rocprofv3 prints for it) or demangled
    m = re.search(r"gemm_dma_kernelI\w*?Li\d+ELi\d+ELi\d+ELi\d+ELi\d+ELi\d+ELb([01])ELb[01]ELb[01]ELb([01])E", name)
    if m:
        conv, xa = m.group(1) == "1", m.group(2) == "1"
        return "conv2d" if conv else ("linear_xattn" if xa else "linear")
    m = re.search(r"gemm_dma_kernel<[^>]*?,\s*(true|false),\s*(?:true|false),\s*(?:true|false),\s*(true|false)>", name)
    if m:
        conv, xa = m.group(1) == "true", m.group(2) == "true"
        return "conv2d" if conv else ("linear_xattn" if xa else "linear")
    if "gemm8p_kernel" in name:
        return "linear"
    if "gemm_kernel" in name:                      # register-staged fallback (ragged K)
        return "linear"
    if "attn16v2_kernel" in name:                  # the 77-token text context (S < 256) is the only user of the 16-row kernel
        return "attention_cross"
    if "attn" in name:
        return "attention_self"
    if name.startswith("gn_") or "gn_stats" in name or "gn_apply" in name or "gn_finalize" in name:
        return "group_norm"
    if "ln_kernel" in name:
        return "layer_norm"
    if "geglu_kernel" in name:
        return "geglu"
    if "euler_kernel" in name or "step_advance" in name or "timestep" in name:
        return "loop"
    if "split_rows_kernel" in name:                # strict mode: the few matrix operands no producer could leave as a split image
        return "split_f32"
    if "spin_kernel" in name:
        return None
    return "torch_glue"


def main():
    src = sys.argv[1]
    rows = list(csv.DictReader(open(src)))
    steps = sum(int(r["Calls"]) for r in rows if "conv_thin_kernel" in r["Name"])
    if steps == 0:
        raise SystemExit("no conv_thin_kernel launches in the profile: cannot count steps")
    fam = {}
    for r in rows:
        f = family(r["Name"])
        if f is None:
            continue
        d = fam.setdefault(f, {"calls": 0, "ns": 0.0})
        d["calls"] += int(r["Calls"])
        d["ns"] += float(r["TotalDurationNs"])
    out = {"_what": "per-family kernel time per denoise step from the rocprofv3 kernel-trace summary of bench.py",
           "source_csv": os.path.basename(src), "steps_in_run": steps, "families": {}}
    total = 0.0
    for f, d in sorted(fam.items(), key=lambda kv: -kv[1]["ns"]):
        ms = d["ns"] / steps / 1e6
        total += ms
        out["families"][f] = {"ms_per_step": round(ms, 4), "launches_per_step": round(d["calls"] / steps, 2),
                              "avg_launch_us": round(d["ns"] / d["calls"] / 1e3, 3)}
    out["kernel_ms_per_step"] = round(total, 3)
    dst = sys.argv[2] if len(sys.argv) > 2 else re.sub(r"_bench_kernel_stats\.csv$", "_families.json", src)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
