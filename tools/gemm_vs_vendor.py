"""Reference point, not a product path: every Linear shape of a denoise step on this library against torch's F.linear
(hipBLASLt) on the same tensors, cold weights (a ring of weight copies larger than the memory-side cache, as in the step).

    python tools/gemm_vs_vendor.py [out.json] [dtype]      ->  profiles/rNN_vendor.json

Per shape and batch: microseconds and TFLOP/s of both, and the ratio vendor / ours (> 1 = this library is faster)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
from tools import op_bench
from tools.op_bench import timeit

out_path = sys.argv[1] if len(sys.argv) > 1 else None
op_bench.dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
rnd = op_bench.rnd
# (M per image, K, N, launches per step at bs=1, epilogue) - the step's Linear census (SURVEY 8a row L after the q|k|v / GEGLU fusions)
SHAPES = [(1024, 1280, 1280, 192, "bias+residual / plain"), (1024, 1280, 3840, 60, "q|k|v"), (1024, 1280, 10240, 60, "GEGLU proj (2 x 5120 rows)"),
          (1024, 5120, 1280, 60, "FF2 + residual"), (4096, 640, 640, 40, "bias+residual / plain"), (4096, 640, 1920, 10, "q|k|v"),
          (4096, 640, 5120, 10, "GEGLU proj (2 x 2560 rows)"), (4096, 2560, 640, 10, "FF2 + residual")]
rows = []
for B in (1, 2, 4):
    tot_o = tot_v = 0.0
    for Mi, K, N, count, what in SHAPES:
        M = Mi * B
        x, b = rnd(M, K), rnd(N)
        ncopy = max(1, min(32, int(600e6 // (N * K * 2))))
        ws = [rnd(N, K) * K ** -0.5 for _ in range(ncopy)]
        it = [0]
        def ours():
            it[0] += 1
            return ops.linear(x, ws[it[0] % ncopy], b)
        def vendor():
            it[0] += 1
            return torch.nn.functional.linear(x, ws[it[0] % ncopy], b)
        uo, uv = timeit(ours, iters=max(20, ncopy)), timeit(vendor, iters=max(20, ncopy))
        fl = 2.0 * M * K * N
        tot_o += uo * count; tot_v += uv * count
        rows.append({"batch": B, "M": M, "K": K, "N": N, "launches_per_step": count, "what": what, "ours_us": round(uo, 2), "vendor_us": round(uv, 2),
                     "ours_tflops": round(fl / uo / 1e6, 1), "vendor_tflops": round(fl / uv / 1e6, 1), "vendor_over_ours": round(uv / uo, 3)})
        print(f"B={B} M={M:6d} K={K:5d} N={N:5d}: ours {uo:7.1f} us {fl/uo/1e6:7.1f} TF/s | hipBLASLt {uv:7.1f} us {fl/uv/1e6:7.1f} TF/s | vendor/ours {uv/uo:5.2f}", flush=True)
    rows.append({"batch": B, "weighted_step_sum_us": {"ours": round(tot_o, 1), "vendor": round(tot_v, 1)}})
    print(f"B={B} launches-weighted sum: ours {tot_o:.0f} us, hipBLASLt {tot_v:.0f} us")
if out_path:
    json.dump({"_what": "Linear shapes of one denoise step: this library (st_linear, bias epilogue, product dispatch) vs torch F.linear (hipBLASLt), "
                        "cold weights, hipGraph-replayed launches; vendor_over_ours > 1 = this library faster", "dtype": str(op_bench.dt),
               "device": torch.cuda.get_device_name(0), "rows": rows}, open(out_path, "w"), indent=1)
