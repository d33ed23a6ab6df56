"""Developer timing of the four-wave 256x256 GEMM (cfg 102) against the eight-phase one (cfg 100) on a few shapes, no checks.
Needs a dev build (ST_VARIANT=<name>, -DST_DEV_CONFIGS).  usage: gemm4w_time.py [cfg ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.op_bench import timeit, rnd  # noqa: E402  (selects the ST_VARIANT build)
from stabletriton_amd import _C, ops  # noqa: E402

lib = _C.load()
force = lib.st_debug_force_gemm
force.argtypes, force.restype = [ctypes.c_int, ctypes.c_int], None
cfgs = [int(c) for c in sys.argv[1:]] or [100, 102]
SHAPES = [(4096, 1280, 3840, "plain"), (4096, 2560, 3840, "plain"), (4096, 5120, 3840, "plain"), (4096, 1280, 3840, "res"),
          (4096, 1280, 5120, "geglu"), (4096, 5120, 1280, "res"), (4096, 4096, 4096, "plain"), (1024, 1280, 5120, "geglu"), (8192, 8192, 8192, "plain")]
for M, K, N, kind in SHAPES:
    rows = 2 * N if kind == "geglu" else N
    x, w, b = rnd(M, K), rnd(rows, K) * K ** -0.5, rnd(rows)
    res = rnd(M, N) if kind == "res" else None
    line = f"M={M:5d} K={K:5d} N={N:5d} {kind:6s}:"
    for cfg in cfgs:
        force(cfg, -1)
        us = timeit(lambda: ops.linear(x, w, b, geglu=(kind == "geglu"), residual=res))
        line += f"  cfg {cfg}: {us:7.1f} us {2.0 * M * K * rows / us / 1e6:7.1f} TF/s"
    force(-1, -1)
    print(line, flush=True)
