"""Developer sweep: every tile configuration x K split of the LDS-DMA GEMM for the Linear shapes of a batch size,
against the cost model's own choice (needs the -DST_DEV_CONFIGS library: ST_VARIANT=dev).
usage: gemm_sweep.py <batch> [geglu|plain|all] [base|refiner]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import _C, ops  # noqa: E402
from tools.op_bench import timeit, rnd  # noqa: E402

lib = _C.load()
force = lib.st_debug_force_gemm
force.argtypes, force.restype = [ctypes.c_int, ctypes.c_int], None
# only configurations the product dispatch can select (the other developer tilings are not maintained: 256x256_W8 faults)
NAMES = {7: "64x64_W8", 8: "128x64_W8", 9: "128x128_W8", 10: "64x128_W8", 19: "256x128_W8", 23: "128x320_W8",
         25: "64x320_W8", 26: "64x80_W4", 27: "128x80_W8", 28: "128x160_W8", 100: "256x256_8P", 101: "256x160_8P", 102: "256x256_4W"}
if os.environ.get("ST_BENCH_DTYPE") == "fp32":      # split operands: two accumulator sets, no eight-phase kernel
    NAMES = {7: "64x64_W8", 8: "128x64_W8", 9: "128x128_W8", 10: "64x128_W8", 27: "128x80_W8", 13: "128x128_W8_S3", 21: "128x64_W8_S3", 22: "64x128_W8_S3",
             15: "64x128_W8_U2", 16: "128x64_W8_U2", 17: "64x64_W8_U2", 20: "128x128_W8_S2"}
ctx = ops.ExecContext()      # (fp32 / strict mode: the split images of the weights are kept per context, as in a compiled module)
ctx.__enter__()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
which = sys.argv[2] if len(sys.argv) > 2 else "all"
shapes = [(1024 * B, 1280, 1280, 0), (1024 * B, 1280, 3840, 0), (1024 * B, 1280, 5120, 1), (1024 * B, 5120, 1280, 0),
          (4096 * B, 640, 640, 0), (4096 * B, 640, 1920, 0), (4096 * B, 640, 2560, 1), (4096 * B, 2560, 640, 0)]
if len(sys.argv) > 3 and sys.argv[3] == "refiner":      # SDXL-refiner (config #5): 768 channels at 64 x 64, 1536 at 32 x 32
    shapes = [(1024 * B, 1536, 1536, 0), (1024 * B, 1536, 4608, 0), (1024 * B, 1536, 6144, 1), (1024 * B, 6144, 1536, 0),
              (4096 * B, 768, 768, 0), (4096 * B, 768, 2304, 0), (4096 * B, 768, 3072, 1), (4096 * B, 3072, 768, 0)]
for M, K, N, geglu in shapes:
    if which == "geglu" and not geglu or which == "plain" and geglu:
        continue
    rows = 2 * N if geglu else N
    x, b, res = rnd(M, K), rnd(rows), rnd(M, N)
    ncopy = max(1, min(32, int(600e6 // (rows * K * x.element_size()))))
    ws = [rnd(rows, K) * K ** -0.5 for _ in range(ncopy)]
    it = [0]

    def call():
        it[0] += 1
        return ops.linear(x, ws[it[0] % ncopy], b, geglu=bool(geglu), residual=None if geglu else res)
    fl = 2.0 * M * K * rows
    force(-1, -1)
    base = timeit(call, iters=max(20, ncopy))
    rows_out = []
    for cfg, name in NAMES.items():
        if geglu and name in ("64x320_W8", "128x80_W8", "64x80_W4"):
            continue
        for sk in (1, 2, 3, 4):
            if cfg >= 100 and sk > 1:
                continue
            force(cfg, sk)
            try:
                us = timeit(call, iters=max(20, ncopy))
            except Exception as e:          # a configuration that rejects the shape
                continue
            rows_out.append((us, name, sk))
    force(-1, -1)
    base2 = timeit(call, iters=max(20, ncopy))          # again, after the sweep: the first timing of a shape reads high
    base = min(base, base2)
    rows_out.sort()
    best = ", ".join(f"{n}/k{sk} {us:.1f}" for us, n, sk in rows_out[:5])
    print(f"M={M:6d} K={K:5d} N={N:5d} g={geglu}: model {base:7.1f} us ({fl / base / 1e6:6.1f} TF/s) | best {best} ({fl / rows_out[0][0] / 1e6:6.1f} TF/s)", flush=True)
