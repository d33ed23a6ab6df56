import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tools.op_bench import timeit, rnd
from stabletriton_amd import ops
B, T, H = 1, 1024, 20
q = rnd(B, T, H * 64)
for S in (256, 384, 512, 768, 1024, 1536, 2048):
    k, v = rnd(B, S, H * 64), rnd(B, S, H * 64)
    print(os.environ.get("ST_VARIANT"), "T=1024 H=20 S=%d" % S, f"{timeit(lambda: ops.attention(q, k, v, H, 0.125)):.2f} us", flush=True)
