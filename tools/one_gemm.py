"""Run one GEMM shape a few times (for rocprofv3 --pmc runs).  usage: one_gemm.py M K N [geglu]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
M, K, N = (int(a) for a in sys.argv[1:4]); geglu = len(sys.argv) > 4
dev = torch.device("cuda:0")
x = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
w = ((torch.rand(2 * N if geglu else N, K, device=dev) * 2 - 1) * K ** -0.5).bfloat16()
for _ in range(5):
    ops.linear(x, w, None, geglu=geglu)
torch.cuda.synchronize()
