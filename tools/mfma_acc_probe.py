"""How does v_mfma_f32_16x16x32_f16 accumulate?  The strict GEMM (split operands) is ~8x noisier against float64 than the exact-fp32
MFMA kernel although the split representation itself is as good as fp32 rounding (CPU check: mean relative error 2.2e-8 both).
Operands that ARE halves (lo = 0) leave only the matrix instruction's own accumulation: signed mean error (truncation shows as a
bias that grows with the number of accumulation steps) and rms, per K, for positive and for zero-mean operands; the exact-fp32
kernel (ops.STRICT_SPLIT = False) beside it."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for kind in ("positive", "zero-mean"):
    for K in (64, 320, 1280, 5120):
        M = N = 256
        a = torch.randn(M, K).half().float()
        b = (torch.randn(N, K) * K ** -0.5).half().float()
        if kind == "positive":
            a, b = a.abs(), b.abs()
        ref = a.double() @ b.double().T
        scale = float(ref.abs().mean())
        row = []
        for split in (True, False):
            ops.STRICT_SPLIT = split
            out = ops.linear(a.to(dev), b.to(dev), None).double().cpu()
            err = (out - ref) / scale
            row.append(f"{'split f16 MFMA' if split else 'exact f32 MFMA'}: mean {float(err.mean()):+.3e} rms {float(err.pow(2).mean().sqrt()):.3e} max {float(err.abs().max()):.3e}")
        ops.STRICT_SPLIT = True
        t = (a @ b.T).double()
        e32 = (t - ref) / scale
        print(f"{kind:9s} K={K:5d} | " + " | ".join(row) + f" | torch CPU fp32: mean {float(e32.mean()):+.3e} rms {float(e32.pow(2).mean().sqrt()):.3e}", flush=True)
