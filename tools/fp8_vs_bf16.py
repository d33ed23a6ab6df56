"""Developer timing: the three projections of the fp8 plan (q|k|v, GEGLU projection, feed-forward output) in e4m3 against their
bf16 forms, same shapes, cold weights, graph-replayed.  usage: python tools/fp8_vs_bf16.py [batch ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.op_bench import timeit, rnd
from stabletriton_amd import ops
batches = [int(b) for b in sys.argv[1:]] or [1, 2, 4]
for B in batches:
    for (Mi, K, N, kind) in ((1024, 1280, 3840, "ln"), (1024, 1280, 5120, "lng"), (1024, 5120, 1280, "res"), (4096, 640, 1920, "ln"), (4096, 640, 2560, "lng"), (4096, 2560, 640, "res")):
        M = Mi * B
        geglu = kind == "lng"
        rows = 2 * N if geglu else N
        ncopy = max(1, min(16, int(400e6 // (rows * K * 2))))
        with ops.ExecContext(hints=False) as ctx:
            x = rnd(M, K)
            ws = [rnd(rows, K) * K ** -0.5 for _ in range(ncopy)]
            b = rnd(rows)
            it = [0]
            if kind != "res":
                g, be = rnd(K), rnd(K)
                eye = rnd(K, K) * K ** -0.5
                xin, st, act = ops.linear(x, eye, None, residual=rnd(M, K), emit_stats=True, emit_q8=("b", 0))
                ctx.fp8.update()
                xin, st, act = ops.linear(x, eye, None, residual=rnd(M, K), emit_stats=True, emit_q8=("b", 0))
                f16 = [ops.fold_layer_norm(g, be, w, b) for w in ws]
                f8 = [ops.fold_layer_norm_fp8(g, be, w, b) for w in ws]
                def bf():
                    it[0] += 1
                    wf, c, d = f16[it[0] % ncopy]
                    return ops.ln_linear(xin, st, wf, c, d, 1e-5, geglu=geglu)
                def f8f():
                    it[0] += 1
                    wq, wsc, c, d = f8[it[0] % ncopy]
                    return ops.linear_fp8x(act, wq, wsc, None, geglu=geglu, ln=(st, c, d, 1e-5), emit_q8=("b", 1) if geglu else None, want_out=not geglu)
            else:
                res = rnd(M, N)
                _, act = ops.linear(x, torch.eye(K, device=x.device, dtype=x.dtype), None, emit_q8=("b", 2))
                ctx.fp8.update()
                _, act = ops.linear(x, torch.eye(K, device=x.device, dtype=x.dtype), None, emit_q8=("b", 2))
                q8 = [ops.quantize_weight_fp8(w) for w in ws]
                def bf():
                    it[0] += 1
                    return ops.linear(x, ws[it[0] % ncopy], b, residual=res, emit_stats=True)
                def f8f():
                    it[0] += 1
                    wq, wsc = q8[it[0] % ncopy]
                    return ops.linear_fp8x(act, wq, wsc, b, residual=res, emit_stats=True, emit_q8=("b", 3))
            ub, u8 = timeit(bf, iters=max(20, ncopy)), timeit(f8f, iters=max(20, ncopy))
            fl = 2.0 * M * K * rows
            print(f"B={B} M={M:6d} K={K:5d} N={N:5d} {kind:4s}: bf16 {ub:7.1f} us {fl/ub/1e6:7.1f} TF/s | e4m3 {u8:7.1f} us {fl/u8/1e6:7.1f} TF/s | x{ub/u8:4.2f}", flush=True)
