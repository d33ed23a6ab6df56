"""Developer timing of the self-attention launch (ST_VARIANT=<name>: a tools/_variants build): the step's shapes, then 1,024 query rows x 20 heads
over key counts 256 ... 2,048 (fixed cost + per-trip slope of the launch)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tools.op_bench import timeit, rnd
from stabletriton_amd import ops
for B, T, H in ((1, 1024, 20), (1, 4096, 10), (2, 1024, 20), (4, 1024, 20), (4, 4096, 10)):
    q, k, v = rnd(B, T, H * 64), rnd(B, T, H * 64), rnd(B, T, H * 64)
    print(os.environ.get("ST_VARIANT"), B, T, H, f"{timeit(lambda: ops.attention(q, k, v, H, 0.125)):.2f} us", flush=True)
B, T, H = 1, 1024, 20
q = rnd(B, T, H * 64)
for S in (256, 512, 1024, 2048):
    k, v = rnd(B, S, H * 64), rnd(B, S, H * 64)
    print(os.environ.get("ST_VARIANT"), "T=1024 H=20 S=%d" % S, f"{timeit(lambda: ops.attention(q, k, v, H, 0.125)):.2f} us", flush=True)
