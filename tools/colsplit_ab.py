"""Developer A/B of the column split (dispatch.h, round 5): the LayerNorm-folded GEGLU projection and a plain projection at the sizes
whose 256 x 256 tiling leaves a last round at most half full.  Needs a -DST_DEV_CONFIGS build: run once with ST_NO_COLSPLIT=1 and once
without (ST_VARIANT=dev)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.op_bench import timeit, rnd
from stabletriton_amd import ops
for M, K, N, geglu in [(4096, 1280, 5120, 1), (2048, 1280, 5120, 1), (4096, 1280, 10240, 0), (8192, 640, 2560, 1), (4096, 1536, 6144, 1), (2048, 1536, 6144, 1)]:
    rows = 2 * N if geglu else N
    x, w, b = rnd(M, K), rnd(rows, K) * K ** -0.5, rnd(rows)
    g, be = rnd(K), rnd(K)
    wf, c, d = ops.fold_layer_norm(g, be, w, b)
    wp, res = rnd(K, K) * K ** -0.5, rnd(M, K)
    xin, st = ops.linear(x, wp, None, residual=res, emit_stats=True)
    ncopy = max(1, min(16, int(600e6 // (rows * K * 2))))
    wfs = [wf.clone() for _ in range(ncopy)]
    it = [0]
    def ln():
        it[0] += 1
        return ops.ln_linear(xin, st, wfs[it[0] % ncopy], c, d, 1e-5, geglu=bool(geglu))
    def plain():
        it[0] += 1
        return ops.linear(x, wfs[it[0] % ncopy], b, geglu=bool(geglu))
    print(f"M={M} K={K} N={N} geglu={geglu} (NO_COLSPLIT={os.environ.get('ST_NO_COLSPLIT', '0')}): ln_linear {timeit(ln, iters=max(20, ncopy)):.1f} us, linear+bias {timeit(plain, iters=max(20, ncopy)):.1f} us", flush=True)
