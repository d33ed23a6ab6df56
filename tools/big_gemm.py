"""GEMM at 4096^3 / 8192x4096x4096 (compare with the CDNA guide's ladder).  ST_GEMM_FORCE selects the config."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
dev = torch.device("cuda:0")
for (M, K, N) in [(4096, 4096, 4096), (8192, 4096, 4096), (16384, 1280, 1280), (4096, 1280, 10240)]:
    x = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * K ** -0.5).bfloat16()
    for _ in range(3):
        ops.linear(x, w, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.linear(x, w, None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={M} K={K} N={N}: {us:8.1f} us {2.0*M*K*N/us/1e6:7.1f} TF/s")
