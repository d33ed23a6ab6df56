"""Developer sweep: tile config x in-launch split-K for the mid-size Linear shapes (needs the dev variant:
tools/build_variant.sh dev -DST_DEV_CONFIGS; run with ST_VARIANT=dev)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops, _C
from tools.op_bench import timeit, rnd
lib = _C.load()
lib.st_debug_force_gemm.argtypes = [C.c_int, C.c_int]
CFG = {"64x64": 7, "128x64": 8, "128x128": 9, "64x128": 10, "256x128": 19}
SHAPES = [(1024, 1280, 1280), (1024, 5120, 1280), (4096, 640, 640), (4096, 2560, 640), (1024, 1280, 3840), (4096, 640, 1920)]
for M, K, N in SHAPES:
    x, w, b, res = rnd(M, K), rnd(N, K) * K ** -0.5, rnd(N), rnd(M, N)
    lib.st_debug_force_gemm(-1, -1)
    ref = ops.linear(x, w, b, residual=res).float()
    base = timeit(lambda: ops.linear(x, w, b, residual=res))
    print(f"M={M} K={K} N={N}: default {base:6.1f} us")
    for name, cfg in CFG.items():
        row = []
        for fk in (1, 2, 3, 4, 6, 8):
            lib.st_debug_force_gemm(cfg, fk)
            out = ops.linear(x, w, b, residual=res).float()
            err = (out - ref).abs().max().item()
            us = timeit(lambda: ops.linear(x, w, b, residual=res))
            row.append(f"sk{fk}:{us:6.1f}" + ("" if err < 0.1 else f"(ERR {err:.2g})"))
        print(f"   {name:8s} " + "  ".join(row))
lib.st_debug_force_gemm(-1, -1)
