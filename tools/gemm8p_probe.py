"""Developer probe of the eight-phase GEMM: cycle stamps (s_memtime) of wave 0 of every block - entry, first K tile landed,
K loop done, stores acknowledged - from a -DST_PROBE8 build:
    tools/build_one_variant.sh probe8 gemm_dense_bf16.hip gemm_api.hip -DST_PROBE8 -DST_DEV_CONFIGS
    ST_VARIANT=probe8 python tools/gemm8p_probe.py [M K N [geglu]] ...
Prints the median / max over blocks of prologue, loop (and per K tile), epilogue in cycles, the launch's span, and its time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.op_bench import timeit, rnd  # noqa: E402  (selects the ST_VARIANT build)
from stabletriton_amd import ops  # noqa: E402

shapes = [(4096, 1280, 3840, 0), (4096, 1280, 5120, 1), (1024, 1280, 5120, 1), (16384, 640, 2560, 1), (16384, 2560, 640, 0)]
if len(sys.argv) >= 4:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)]
if os.environ.get("ST_PROBE_CFG"):          # force a tile configuration (100 = 256 x 256, 101 = 256 x 160): how does the loop's pace depend on how many CUs pull?
    import ctypes
    from stabletriton_amd import _C
    _f = _C.load().st_debug_force_gemm
    _f.argtypes, _f.restype = [ctypes.c_int, ctypes.c_int], None
    _f(int(os.environ["ST_PROBE_CFG"]), -1)
    shapes = [(256, 1280, 3840, 0), (1024, 1280, 3840, 0), (2048, 1280, 3840, 0), (4096, 1280, 3840, 0), (4096, 1280, 7680, 0), (4096, 5120, 3840, 0)]
ctx = ops.ExecContext()
with ctx:
    for M, K, N, geglu in shapes:
        rows = 2 * N if geglu else N
        x, w, b = rnd(M, K), rnd(rows, K) * K ** -0.5, rnd(rows)
        ws = ctx.gemm_workspace(x.device)
        for _ in range(3):
            ops.linear(x, w, b, geglu=bool(geglu))
        torch.cuda.synchronize()
        ws[65536:65536 + 64 * 4096].zero_()
        for _ in range(8):               # back to back, as in the step (the clock settles): the stamps are the last launch's
            ops.linear(x, w, b, geglu=bool(geglu))
        torch.cuda.synchronize()
        st = ws[65536:65536 + 64 * 4096].view(torch.int64).view(-1, 8).cpu()
        st = st[st[:, 0] != 0]
        if st.numel() == 0:
            print(f"M={M} K={K} N={N} g={geglu}: no stamps (not on the eight-phase kernel, or not a probe build)")
            continue
        pro, loop, epi = (st[:, 1] - st[:, 0]).float(), (st[:, 2] - st[:, 1]).float(), (st[:, 3] - st[:, 2]).float()
        span = int(st[:, 3].max() - st[:, 0].min())
        real_us = (st[:, 5] - st[:, 4]).float() / 100.0                  # s_memrealtime: 100 MHz
        ghz = ((st[:, 3] - st[:, 0]).float() / real_us / 1e3).median()
        span_us = float(st[:, 5].max() - st[:, 4].min()) / 100.0
        us = timeit(lambda: ops.linear(x, w, b, geglu=bool(geglu)), iters=20)
        ent = (st[:, 4] - st[:, 4].min()).float() / 100.0                # when the blocks start / end, against the first entry (us)
        ext = (st[:, 5] - st[:, 4].min()).float() / 100.0
        q = lambda v, f: float(v.sort().values[min(int(f * (v.numel() - 1) + 0.5), v.numel() - 1)])
        ramp = " ".join(f"{q(ent, f):.1f}" for f in (0.1, 0.5, 0.9, 1.0)) + " | exits " + " ".join(f"{q(ext, f):.1f}" for f in (0.0, 0.1, 0.5, 0.9, 1.0))
        nk = K // (32 if os.environ.get("ST_BENCH_DTYPE") == "fp32" else 64)      # (split fp32 operands: K tiles of 32)
        print(f"M={M} K={K} N={N} g={geglu}: {st.shape[0]} blocks | prologue med {pro.median():.0f} max {pro.max():.0f} | loop med {loop.median():.0f} "
              f"({loop.median() / nk:.0f} / K tile) max {loop.max():.0f} | epilogue med {epi.median():.0f} max {epi.max():.0f} | block life med {real_us.median():.1f} us at {ghz:.3f} GHz | "
              f"first entry -> last exit {span_us:.1f} us (entries at 10/50/90/100 %: {ramp}) | {us:.1f} us per launch",
              flush=True)
