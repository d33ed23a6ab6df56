"""Times st_attention at the head sizes of the generic kernel (csrc/attention_anyd.hip) beside head_dim 64 (the tuned kernels),
same channel count and token count: what the coverage kernel costs.  usage: python tools/attn_anyd_bench.py [tokens] [channels]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import os                                  # noqa: E402
from tools.devlib import use_variant       # noqa: E402
use_variant(os.environ.get("ST_VARIANT"))  # (ST_VARIANT=<dev build> ST_ATT_ANYD=1: head_dim 64 on the generic kernel as well)
from stabletriton_amd import ops          # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C = int(sys.argv[2]) if len(sys.argv) > 2 else 640
dev = torch.device("cuda:0")
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    for D in (16, 32, 64, 128):
        H = C // D
        q, k, v = (torch.randn(1, T, C, device=dev, dtype=dtype) for _ in range(3))
        for _ in range(3):
            ops.attention(q, k, v, H, D ** -0.5)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.attention(q, k, v, H, D ** -0.5)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1000 / 20
        tf = 4.0 * T * T * C / us / 1e6
        print(f"{str(dtype):15s} head_dim {D:3d} heads {H:3d}: {us:8.1f} us  {tf:7.1f} TFLOP/s")
