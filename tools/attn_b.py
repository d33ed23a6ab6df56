"""Developer timing of the self-attention shapes at several batch sizes (graph-replayed, like tools/op_bench.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
from tools.op_bench import timeit, rnd
for B in (1, 2, 4):
    for T, H in ((4096, 10), (1024, 20)):
        q, k, v = rnd(B, T, H * 64), rnd(B, T, H * 64), rnd(B, T, H * 64)
        us = timeit(lambda: ops.attention(q, k, v, H, 0.125))
        print(f"B={B} T=S={T} H={H}: {us:8.1f} us  {4.0 * B * H * T * T * 64 / us / 1e6:7.1f} TF/s", flush=True)
