"""Developer timing: 3x3 convs of the SDXL-refiner that SDXL-base's tuning never met - the 16 x 16 level (1536 -> 1536, 3072 -> 1536
channels) and the 128 x 128 level (384 -> 384, 768 -> 384) - at batch 1 / 2 / 4; ST_VARIANT=<name> times a tools/_variants build
instead of the product (round 4: the halo kernel against the implicit-GEMM path; 128- against 160-channel tiles)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.op_bench import timeit, rnd
from stabletriton_amd import ops
cl = torch.channels_last
for N in (1, 2, 4):
    for Cin, Cout, H in ((1536, 1536, 16), (3072, 1536, 16), (384, 384, 128), (768, 384, 128)):
        x = rnd(N, Cin, H, H).contiguous(memory_format=cl)
        w = (rnd(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5).contiguous(memory_format=cl)
        b = rnd(Cout)
        print(os.environ.get("ST_VARIANT", "product"), N, Cin, Cout, H, round(timeit(lambda: ops.conv2d(x, w, b, 1, 1)), 1), "us", flush=True)
