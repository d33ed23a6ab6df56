"""Developer A/B: the strict mode's Linear shapes on the eight-phase kernel (ST_GEMM_FORCE=101) against the single-phase 128 x 160
tile (28) and the dispatch's own choice (unset); one process per setting (the knob is read once).
    python tools/strict8p_ab.py            (spawns itself three times)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from tools.op_bench import timeit, rnd
    from stabletriton_amd import ops
    shapes = [(1024, 1280, 5120, 1), (4096, 640, 2560, 1), (2048, 1280, 5120, 1), (4096, 1280, 5120, 1), (4096, 1280, 3840, 0), (4096, 5120, 1280, 0),
              (1024, 1280, 3840, 0), (16384, 640, 2560, 1), (16384, 2560, 640, 0)]
    ctx = ops.ExecContext()                     # (keeps the weights' split images: outside a context every call splits W again)
    with ctx:
        for M, K, N, g in shapes:
            rows = 2 * N if g else N
            x, w, b = rnd(M, K).float(), (rnd(rows, K) * K ** -0.5).float(), rnd(rows).float()
            try:
                us = timeit(lambda: ops.linear(x, w, b, geglu=bool(g)), iters=20)
                us_split = timeit(lambda: ops.split_rows(x), iters=20)          # the activation's image, made per call here (in the step: by the producer)
                print(f"  M={M} K={K} N={N} geglu={g}: {us:7.1f} us, of which split_rows(x) {us_split:5.1f}", flush=True)
            except Exception as e:
                print(f"  M={M} K={K} N={N} geglu={g}: {type(e).__name__} {e}", flush=True)
else:
    for force in ("", "101", "28"):
        env = dict(os.environ, ST_BENCH_DTYPE="fp32")
        if force:
            env["ST_GEMM_FORCE"] = force
        print(f"ST_GEMM_FORCE={force or '(dispatch)'}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
