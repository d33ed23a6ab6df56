"""Developer sweep: the implicit-GEMM convs of the step (1x1 shortcuts, also over a never-written concatenation; stride-2
downsamplers) on every tile configuration x K split against the cost model's choice (needs a -DST_DEV_CONFIGS build of
gemm_api / gemm_conv_bf16: ST_VARIANT=<name>).  usage: python tools/conv_cfg_sweep.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.op_bench import timeit, rnd
from stabletriton_amd import _C, ops
force = _C.load().st_debug_force_gemm
force.argtypes, force.restype = [ctypes.c_int, ctypes.c_int], None
NAMES = {7: "64x64", 8: "128x64", 9: "128x128", 10: "64x128", 19: "256x128", 27: "128x80", 28: "128x160"}
cl = torch.channels_last
# (C0, C1, H, Cout, k, stride): C1 > 0 = two-source 1x1
SHAPES = [(1280, 1280, 32, 1280, 1, 1), (1280, 640, 32, 1280, 1, 1), (1280, 640, 64, 640, 1, 1), (640, 640, 64, 640, 1, 1), (640, 320, 64, 640, 1, 1),
          (640, 320, 128, 320, 1, 1), (320, 320, 128, 320, 1, 1), (640, 0, 32, 1280, 1, 1), (320, 0, 64, 640, 1, 1),
          (320, 0, 128, 320, 3, 2), (640, 0, 64, 640, 3, 2)]
for C0, C1, H, Cout, k, stride in SHAPES:
    Cin = C0 + C1
    a = rnd(1, C0, H, H).contiguous(memory_format=cl)
    b = rnd(1, C1, H, H).contiguous(memory_format=cl) if C1 else None
    w = (rnd(Cout, Cin, k, k) * (Cin * k * k) ** -0.5).contiguous(memory_format=cl)
    bias = rnd(Cout)
    Ho = H // stride
    res = rnd(1, Cout, Ho, Ho).contiguous(memory_format=cl)
    call = (lambda: ops.conv2d_cat(a, b, w, bias, residual=res, emit_colstats=True)) if C1 else (lambda: ops.conv2d(a, w, bias, stride, k // 2, residual=res, emit_colstats=True))
    force(-1, -1)
    base = timeit(call)
    rows = []
    for cfg, name in NAMES.items():
        for sk in (1, 2, 3, 4, 6):
            force(cfg, sk)
            try:
                rows.append((timeit(call), name, sk))
            except Exception:
                pass
    force(-1, -1)
    base = min(base, timeit(call))
    rows.sort()
    fl = 2.0 * Ho * Ho * Cout * Cin * k * k
    print(f"Cin={C0}+{C1} H={H} Cout={Cout} k={k} s={stride}: model {base:6.1f} us {fl / base / 1e6:6.1f} TF/s | best " + ", ".join(f"{n}/k{s} {u:.1f}" for u, n, s in rows[:5]), flush=True)
