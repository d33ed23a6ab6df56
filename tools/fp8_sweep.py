"""Developer sweep: the three projections of the fp8 plan (q|k|v, GEGLU projection, feed-forward output) in e4m3 on every tile
configuration x K split against the cost model's choice (needs a -DST_DEV_CONFIGS build of gemm_api / gemm_fp8 / gemm_4w:
ST_VARIANT=<name>).  usage: python tools/fp8_sweep.py [batch ...] [refiner]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.op_bench import timeit, rnd
from stabletriton_amd import _C, ops
force = _C.load().st_debug_force_gemm
force.argtypes, force.restype = [ctypes.c_int, ctypes.c_int], None
NAMES = {7: "64x64", 8: "128x64", 9: "128x128", 10: "64x128", 19: "256x128", 27: "128x80", 28: "128x160", 100: "8p-256", 101: "8p-160"}
refiner = "refiner" in sys.argv[1:]              # SDXL-refiner's projections (config #5) instead of SDXL-base's
batches = [int(b) for b in sys.argv[1:] if b != "refiner"] or [1, 2, 4]
BASE = ((1024, 1280, 3840, "ln"), (1024, 1280, 5120, "lng"), (1024, 5120, 1280, "res"), (4096, 640, 1920, "ln"), (4096, 640, 2560, "lng"), (4096, 2560, 640, "res"))
REFI = ((1024, 1536, 4608, "ln"), (1024, 1536, 6144, "lng"), (1024, 6144, 1536, "res"), (4096, 768, 2304, "ln"), (4096, 768, 3072, "lng"), (4096, 3072, 768, "res"))
for B in batches:
    for (Mi, K, N, kind) in (REFI if refiner else BASE):
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
            force(-1, -1)
            base = timeit(f8f, iters=max(20, ncopy))
            rows_out = []
            for cfg, name in NAMES.items():
                if geglu and name == "128x80":
                    continue
                for sk in ((1, 2, 3, 4) if kind == "res" and cfg < 100 else (1,)):
                    force(cfg, sk)
                    try:
                        rows_out.append((timeit(f8f, iters=max(20, ncopy)), name, sk))
                    except Exception:
                        pass
            force(-1, -1)
            base = min(base, timeit(f8f, iters=max(20, ncopy)))
            rows_out.sort()
            fl = 2.0 * M * K * rows
            print(f"B={B} M={M:6d} K={K:5d} N={N:5d} {kind:4s}: model {base:7.1f} us {fl / base / 1e6:7.1f} TF/s | best " +
                  ", ".join(f"{n}/k{k} {u:.1f}" for u, n, k in rows_out[:5]), flush=True)
