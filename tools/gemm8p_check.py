"""Developer check of the 256-row GEMM kernels (eight-phase 256x256 / 256x160 = cfg 100 / 101, four-wave 256x256 = cfg 102) against torch (fp32 reference on the bf16-rounded operands) and A/B
timing against the cost model's choice.  Needs the dev library:  tools/build_variant.sh dev -DST_DEV_CONFIGS  and
ST_VARIANT=dev."""
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
dev = torch.device("cuda:0")


def ref_linear(x, w, b, geglu, res):
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    if geglu:
        a, g = y.chunk(2, -1)
        y = a * torch.nn.functional.gelu(g)
    if res is not None:
        y = y + res.float()
    return y


def check(M, K, N, geglu=False, ln=False, residual=False, stats=False):
    rows = 2 * N if geglu else N
    x, w, b = rnd(M, K), rnd(rows, K) * K ** -0.5, rnd(rows)
    res = rnd(M, N) if residual else None
    out = {}
    for cfg in (100, 101, 102, -1):
        force(cfg, -1)
        if ln:
            g, be = rnd(K) * 0.1 + 1.0, rnd(K) * 0.1
            wf, c, d = ops.fold_layer_norm(g, be, w, b)
            wp = rnd(K, K) * K ** -0.5
            force(-1, -1)
            _, st = ops.linear(x, wp, None, residual=rnd(M, K), emit_stats=True)
            xin = _
            force(cfg, -1)
            fn = lambda: ops.ln_linear(xin, st, wf, c, d, 1e-5, geglu=geglu)
            xn = torch.nn.functional.layer_norm(xin.float(), (K,), g.float(), be.float(), 1e-5)
            ref = ref_linear(xn, w, b, geglu, None)
        elif stats:
            fn = lambda: ops.linear(x, w, b, residual=res, emit_stats=True)
            ref = ref_linear(x, w, b, geglu, res)
        else:
            fn = lambda: ops.linear(x, w, b, geglu=geglu, residual=res)
            ref = ref_linear(x, w, b, geglu, res)
        y = fn()
        extra = ""
        if stats:
            y, st = y
            s = st.buf.float().sum(1)
            yy = y.float()
            e1 = float((s[:, 0] - yy.sum(1)).abs().max() / yy.sum(1).abs().max())
            e2 = float((s[:, 1] - (yy * yy).sum(1)).abs().max() / (yy * yy).sum(1).abs().max())
            extra = f" stats err {e1:.1e}/{e2:.1e} chunks {st.chunks}"
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        us = timeit(fn)
        out[cfg] = (err, us, extra)
    fl = 2.0 * M * K * rows
    names = {100: "8p-256", 101: "8p-160", 102: "4w-256", -1: "model"}
    print(f"M={M:6d} K={K:5d} N={N:5d} geglu={int(geglu)} ln={int(ln)} res={int(residual)} stats={int(stats)}: " +
          " | ".join(f"{names[c]} err {out[c][0]:.2e} {out[c][1]:7.1f} us {fl / out[c][1] / 1e6:7.1f} TF/s" for c in out) +
          "".join(out[c][2] for c in out), flush=True)
    assert all(out[c][0] < 2e-2 for c in out), "a 256-row kernel's result is wrong"
    force(-1, -1)


if __name__ == "__main__":
    torch.manual_seed(0)
    for M in (1024, 2048, 4096):
        check(M, 1280, 5120, geglu=True, ln=True)
        check(M, 1280, 3840, ln=True)
        check(M, 5120, 1280, residual=True, stats=True)
        check(M, 1280, 1280, residual=True, stats=True)
        check(M, 1280, 1280, ln=True)
        check(M, 1280, 1280)
    for M in (4096, 8192, 16384):
        check(M, 640, 2560, geglu=True, ln=True)
        check(M, 2560, 640 if False else 768 if False else 1280, residual=True)      # 256-multiple stand-in for N=640
    check(4096, 640, 1920, ln=True)
    check(4096, 640, 640, residual=True, stats=True)
    check(4096, 4096, 4096)
