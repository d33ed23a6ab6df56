"""Developer sweep: pipeline depth / tile variants of the LDS-DMA GEMM (dev variant: -DST_DEV_CONFIGS, optionally
-DST_FILL_ONLY to time the DMA stream alone).  ST_VARIANT selects the build."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops, _C
from tools.op_bench import timeit, rnd
lib = _C.load()
lib.st_debug_force_gemm.argtypes = [C.c_int, C.c_int]
CFG = {"64x64_S4": 7, "64x64_S8": 14, "64x64_U2": 17, "128x64_S3": 21, "128x64_S4": 8, "128x64_S6": 12, "128x64_U2": 16,
       "64x128_S3": 22, "64x128_S4": 10, "64x128_S6": 11, "64x128_U2": 15, "128x128_S2": 20, "128x128_S3": 9, "128x128_S4": 13,
       "256x128_S3": 19, "64x80_W4": 26, "128x80_W8": 27}
SHAPES = [(1024, 1280, 1280), (1024, 5120, 1280), (4096, 640, 640), (4096, 2560, 640)]
for M, K, N in SHAPES:
    x, w, b = rnd(M, K), rnd(N, K) * K ** -0.5, rnd(N)
    lib.st_debug_force_gemm(-1, -1)
    base = timeit(lambda: ops.linear(x, w, b))
    row = [f"default {base:5.1f}"]
    for name, cfg in CFG.items():
        lib.st_debug_force_gemm(cfg, 1)
        row.append(f"{name} {timeit(lambda: ops.linear(x, w, b)):5.1f}")
    print(f"M={M} K={K} N={N}: " + " | ".join(row))
lib.st_debug_force_gemm(-1, -1)
