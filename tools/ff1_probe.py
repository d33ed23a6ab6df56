"""Developer experiment: the GEGLU projection shape with and without the GEGLU epilogue, across tiles (dev variant)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops, _C
from tools.op_bench import timeit, rnd
lib = _C.load()
lib.st_debug_force_gemm.argtypes = [C.c_int, C.c_int]
CFG = {"128x320": 23, "128x160": 28, "128x128": 9, "256x128": 19, "64x128": 10}
for M, K, N in [(1024, 1280, 5120), (4096, 640, 2560)]:
    x = rnd(M, K)
    w = rnd(2 * N, K) * K ** -0.5
    b = rnd(2 * N)
    for name, cfg in CFG.items():
        lib.st_debug_force_gemm(cfg, 1)
        t_g = timeit(lambda: ops.linear(x, w, b, geglu=True))
        t_p = timeit(lambda: ops.linear(x, w, b))            # plain GEMM on the same 2N x K weights
        t_n = timeit(lambda: ops.linear(x, w, None))
        print(f"M={M} K={K} 2N={2*N} {name:8s}: geglu {t_g:6.1f} us | plain+bias {t_p:6.1f} us | plain {t_n:6.1f} us")
lib.st_debug_force_gemm(-1, -1)
