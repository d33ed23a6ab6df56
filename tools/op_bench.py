"""Micro-benchmark of the HIP operators at the SDXL-base shapes (developer tool).
Usage: python tools/op_bench.py [linear|conv|attn|norm|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops  # noqa: E402
from tools.devlib import use_variant  # noqa: E402

use_variant(os.environ.get("ST_VARIANT"))      # ST_VARIANT=<name>: a tools/_variants/<name> build (developer A/B runs)

dev = torch.device("cuda:0")
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[os.environ.get("ST_BENCH_DTYPE", "bf16")]      # fp32 = the strict mode (split operands)


def timeit(fn, iters=20, warm=3):
    """GPU time per call in us: `iters` calls captured in one hipGraph, replayed (no host launch cost)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


def rnd(*shape):
    return (torch.rand(*shape, device=dev, dtype=torch.float32) * 2 - 1).to(dt)


def linear():
    # (M, K, N, count per step, geglu)
    shapes = [(1024, 1280, 1280, 372, 0), (1024, 1280, 10240 // 2, 60, 1), (1024, 5120, 1280, 60, 0),
              (4096, 640, 640, 70, 0), (4096, 640, 5120 // 2, 10, 1), (4096, 2560, 640, 10, 0),
              (77, 2048, 1280, 120, 0), (77, 2048, 640, 20, 0), (1, 1280, 1280, 9, 0), (1024, 1280, 3840, 0, 0)]
    tot = 0.0
    cold = os.environ.get("COLD", "1") == "1"
    for M, K, N, cnt, geglu in shapes:
        x, b = rnd(M, K), rnd(2 * N if geglu else N)
        nrows = 2 * N if geglu else N
        ncopy = max(1, min(64, int(600e6 // (nrows * K * 2)))) if cold else 1     # > 256 MiB of weights: no cache reuse
        ws = [rnd(nrows, K) * K ** -0.5 for _ in range(ncopy)]
        it = [0]

        def call():
            it[0] += 1
            return ops.linear(x, ws[it[0] % ncopy], b, geglu=bool(geglu))
        us = timeit(call, iters=max(20, ncopy))
        fl = 2.0 * M * K * (2 * N if geglu else N)
        tot += us * cnt
        print(f"linear M={M:5d} K={K:5d} N={N:5d} geglu={geglu}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  x{cnt} = {us * cnt / 1e3:6.2f} ms")
    print(f"linear total per step: {tot / 1e3:.2f} ms")


def lnlinear():
    for M, K, N, geglu in [(1024, 1280, 3840, 0), (1024, 1280, 1280, 0), (1024, 1280, 5120, 1), (4096, 640, 1920, 0), (4096, 640, 2560, 1)]:
        rows = 2 * N if geglu else N
        x, w, b, g, be = rnd(M, K), rnd(rows, K) * K ** -0.5, rnd(rows), rnd(K), rnd(K)
        wf, c, d = ops.fold_layer_norm(g, be, w, b)
        wp, res = rnd(K, K) * K ** -0.5, rnd(M, K)
        t_ln = timeit(lambda: ops.layer_norm(x, g, be, 1e-5))
        t_lin = timeit(lambda: ops.linear(x, w, b, geglu=bool(geglu)))
        t_p0 = timeit(lambda: ops.linear(x, wp, None, residual=res))
        t_p1 = timeit(lambda: ops.linear(x, wp, None, residual=res, emit_stats=True))
        _, st = ops.linear(x, wp, None, residual=res, emit_stats=True)
        t_f = timeit(lambda: ops.ln_linear(x, st, wf, c, d, 1e-5, geglu=bool(geglu)))
        print(f"M={M} K={K} N={N} geglu={geglu}: layer_norm {t_ln:6.1f} + linear {t_lin:6.1f} = {t_ln + t_lin:6.1f} us | folded {t_f:6.1f} us"
              f" + producer stats {t_p1 - t_p0:+5.1f} us (producer {t_p0:5.1f} -> {t_p1:5.1f})")


def conv():
    # (Cin, H, Cout, k, stride, ups, count)
    shapes = [(320, 128, 320, 3, 1, 0, 7), (640, 64, 640, 3, 1, 0, 6), (1280, 32, 1280, 3, 1, 0, 10), (2560, 32, 1280, 3, 1, 0, 2),
              (1920, 32, 1280, 3, 1, 0, 1), (1920, 64, 640, 3, 1, 0, 1), (1280, 64, 640, 3, 1, 0, 1), (960, 64, 640, 3, 1, 0, 1),
              (960, 128, 320, 3, 1, 0, 1), (640, 128, 320, 3, 1, 0, 2), (320, 64, 640, 3, 1, 0, 1), (640, 32, 1280, 3, 1, 0, 1),
              (320, 128, 320, 3, 2, 0, 1), (640, 64, 640, 3, 2, 0, 1), (1280, 32, 1280, 3, 1, 1, 1), (640, 64, 640, 3, 1, 1, 1),
              (2560, 32, 1280, 1, 1, 0, 2), (960, 128, 320, 1, 1, 0, 1), (320, 128, 4, 3, 1, 0, 1), (4, 128, 320, 3, 1, 0, 1)]
    tot = 0.0
    cl = torch.channels_last
    for Cin, H, Cout, k, st, ups, cnt in shapes:
        x = rnd(1, Cin, H, H).contiguous(memory_format=cl)
        w = (rnd(Cout, Cin, k, k) * (Cin * k * k) ** -0.5).contiguous(memory_format=cl)
        b = rnd(Cout)
        us = timeit(lambda: ops.conv2d(x, w, b, st, k // 2, upsample2x=bool(ups)))
        Ho = (H * (2 if ups else 1) + 2 * (k // 2) - k) // st + 1
        fl = 2.0 * Ho * Ho * Cout * Cin * k * k
        tot += us * cnt
        print(f"conv Cin={Cin:5d} H={H:4d} Cout={Cout:5d} k={k} s={st} ups={ups}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s x{cnt} = {us * cnt / 1e3:6.2f} ms")
    print(f"conv total per step: {tot / 1e3:.2f} ms")


def attn():
    tot = 0.0
    for T, S, H, cnt in [(4096, 4096, 10, 10), (1024, 1024, 20, 60), (4096, 77, 10, 10), (1024, 77, 20, 60)]:
        q, k, v = rnd(1, T, H * 64), rnd(1, S, H * 64), rnd(1, S, H * 64)
        us = timeit(lambda: ops.attention(q, k, v, H, 0.125))
        fl = 4.0 * H * T * S * 64
        tot += us * cnt
        print(f"attn T={T:5d} S={S:5d} H={H:3d}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s x{cnt} = {us * cnt / 1e3:6.2f} ms")
    print(f"attention total per step: {tot / 1e3:.2f} ms")


def norm():
    cl = torch.channels_last
    tot = 0.0
    for C, H, cnt in [(320, 128, 8), (640, 128, 2), (960, 128, 1), (320, 64, 1), (640, 64, 11), (960, 64, 1), (1280, 64, 1),
                      (1920, 64, 1), (640, 32, 1), (1280, 32, 16), (1920, 32, 1), (2560, 32, 2)]:
        x = rnd(1, C, H, H).contiguous(memory_format=cl)
        w, b = rnd(C), rnd(C)
        us = timeit(lambda: ops.group_norm(x, 32, w, b, 1e-5, True))
        by = 2.0 * x.numel() * 2
        tot += us * cnt
        print(f"group_norm C={C:5d} H={H:4d}: {us:7.1f} us {by / us / 1e3:7.1f} GB/s x{cnt} = {us * cnt / 1e3:6.2f} ms")
    print(f"group_norm total per step: {tot / 1e3:.2f} ms")
    for M, C, cnt in [(1024, 1280, 180), (4096, 640, 30)]:
        x, w, b = rnd(M, C), rnd(C), rnd(C)
        us = timeit(lambda: ops.layer_norm(x, w, b, 1e-5))
        print(f"layer_norm M={M} C={C}: {us:7.1f} us {2.0 * x.numel() * 2 / us / 1e3:7.1f} GB/s x{cnt} = {us * cnt / 1e3:6.2f} ms")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    for name, fn in (("linear", linear), ("lnlinear", lnlinear), ("conv", conv), ("attn", attn), ("norm", norm)):
        if what in (name, "all"):
            fn()
