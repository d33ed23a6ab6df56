"""Run ONE operator shape a few times (target of the rocprofv3 --pmc passes in tools/pmc_ops.sh).
usage: one_op.py linear M K N [g|ln|lng] | xattn B T C H S | attn B T S H | conv N Cin H Cout k stride ups | gn N C H silu"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[os.environ.get("ST_BENCH_DTYPE", "bf16")]      # fp32 = the strict mode
REPS = 5


def rnd(*shape):
    return (torch.rand(*shape, device=dev, dtype=torch.float32) * 2 - 1).to(dt)


_ctx = ops.ExecContext()      # (strict mode: the split images of the weights are kept per context, as in a compiled module)
_ctx.__enter__()
kind, a = sys.argv[1], sys.argv[2:]
if kind == "linear":
    M, K, N = (int(v) for v in a[:3])
    mode = a[3] if len(a) > 3 else ""
    geglu = "g" in mode
    rows = 2 * N if geglu else N
    x, w, b = rnd(M, K), rnd(rows, K) * K ** -0.5, rnd(rows)
    if "ln" in mode:
        g, be = rnd(K), rnd(K)
        wf, c, d = ops.fold_layer_norm(g, be, w, b)
        wp, res = rnd(K, K) * K ** -0.5, rnd(M, K)
        _, st = ops.linear(x, wp, None, residual=res, emit_stats=True)
        fn = lambda: ops.ln_linear(x, st, wf, c, d, 1e-5, geglu=geglu)
    else:
        fn = lambda: ops.linear(x, w, b, geglu=geglu)
elif kind == "xattn":                  # the cross-attention query projection with the attention in its epilogue
    B, T, C, H, S = (int(v) for v in a[:5])
    x, w, b = rnd(B, T, C), rnd(C, C) * C ** -0.5, rnd(C)
    g, be = rnd(C), rnd(C)
    wf, c, d = ops.fold_layer_norm(g, be, w, b)
    wp, res = rnd(C, C) * C ** -0.5, rnd(B, T, C)
    xin, st = ops.linear(x, wp, None, residual=res, emit_stats=True)
    kv = rnd(B, S, 2 * C)
    k, v = kv[..., :C], kv[..., C:]
    fn = lambda: ops.ln_linear_xattn(xin, st, wf, c, d, 1e-5, k, v, H, 0.125)
elif kind == "attn":
    B, T, S, H = (int(v) for v in a[:4])
    q, k, v = rnd(B, T, H * 64), rnd(B, S, H * 64), rnd(B, S, H * 64)
    fn = lambda: ops.attention(q, k, v, H, 0.125)
elif kind == "conv":
    N, Cin, H, Cout, k, st, ups = (int(v) for v in a[:7])
    cl = torch.channels_last
    x = rnd(N, Cin, H, H).contiguous(memory_format=cl)
    w = (rnd(Cout, Cin, k, k) * (Cin * k * k) ** -0.5).contiguous(memory_format=cl)
    b = rnd(Cout)
    fn = lambda: ops.conv2d(x, w, b, st, k // 2, upsample2x=bool(ups))
elif kind == "gn":
    N, C, H, silu = (int(v) for v in a[:4])
    x = rnd(N, C, H, H).contiguous(memory_format=torch.channels_last)
    w, b = rnd(C), rnd(C)
    fn = lambda: ops.group_norm(x, 32, w, b, 1e-5, bool(silu))
else:
    raise SystemExit(f"unknown op {kind}")
for _ in range(REPS):
    fn()
torch.cuda.synchronize()
