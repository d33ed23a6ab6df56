"""Developer A/B of single operator shapes on ONE box: the product library against tools/_variants/<name> builds.
usage: python tools/ab_ops.py variantA variantB ...     ('product' = the in-tree build); each variant runs in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("linear", 1024, 1280, 1280, "res+stats"), ("linear", 1024, 1280, 1280, "ln"), ("linear", 1024, 1280, 3840, "ln"), ("linear", 1024, 1280, 5120, "lng"),
          ("linear", 1024, 5120, 1280, "res+stats"), ("linear", 4096, 640, 640, "res+stats"), ("linear", 4096, 640, 1920, "ln"), ("linear", 4096, 640, 2560, "lng"),
          ("linear", 4096, 2560, 640, "res+stats"), ("linear", 2048, 1280, 5120, "lng"), ("linear", 4096, 1280, 5120, "lng"), ("linear", 4096, 1280, 3840, "ln"),
          ("conv", 1280, 32, 1280), ("conv", 640, 64, 640), ("conv", 320, 128, 320)]


def one(variant):
    import torch
    from tools.devlib import use_variant
    use_variant(None if variant == "product" else variant)
    from stabletriton_amd import ops
    from tools.op_bench import timeit, rnd
    for sh in SHAPES:
        if sh[0] == "linear":
            _, M, K, N, mode = sh
            geglu = "g" in mode
            rows = 2 * N if geglu else N
            ncopy = max(1, min(16, int(400e6 // (rows * K * 2))))
            x, b = rnd(M, K), rnd(rows)
            ws = [rnd(rows, K) * K ** -0.5 for _ in range(ncopy)]
            it = [0]
            if "ln" in mode:
                g, be = rnd(K), rnd(K)
                folded = [ops.fold_layer_norm(g, be, w, b) for w in ws]
                xin, st = ops.linear(x, rnd(K, K) * K ** -0.5, None, residual=rnd(M, K), emit_stats=True)
                def fn():
                    it[0] += 1
                    wf, c, d = folded[it[0] % ncopy]
                    return ops.ln_linear(xin, st, wf, c, d, 1e-5, geglu=geglu)
            else:
                res = rnd(M, N)
                def fn():
                    it[0] += 1
                    return ops.linear(x, ws[it[0] % ncopy], b, residual=res, emit_stats=True)
            us = timeit(fn, iters=max(20, ncopy))
            print(f"RESULT linear_{M}x{K}x{N}_{mode} {us:.2f}", flush=True)
        else:
            _, C, H, Co = sh
            cl = torch.channels_last
            x = rnd(1, C, H, H).contiguous(memory_format=cl)
            ws = [(rnd(Co, C, 3, 3) * (9 * C) ** -0.5).contiguous(memory_format=cl) for _ in range(8)]
            b = rnd(Co)
            it = [0]
            def fn():
                it[0] += 1
                return ops.conv2d(x, ws[it[0] % 8], b, 1, 1, emit_colstats=True)
            us = timeit(fn, iters=24)
            print(f"RESULT conv_{C}_{H}_{Co} {us:.2f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2]); sys.exit(0)
    variants = sys.argv[1:]
    table = {}
    for rnd_ in range(2):
        for v in variants:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", v], capture_output=True, text=True)
            for l in out.stdout.splitlines():
                if l.startswith("RESULT"):
                    _, name, us = l.split()
                    table.setdefault(name, {}).setdefault(v, []).append(float(us))
            if "RESULT" not in out.stdout:
                print(v, "FAILED", out.stderr[-600:])
    print(f"{'shape':34s}" + "".join(f"{v:>12s}" for v in variants))
    for name, d in table.items():
        print(f"{name:34s}" + "".join(f"{min(d.get(v, [float('nan')])):12.1f}" for v in variants))
