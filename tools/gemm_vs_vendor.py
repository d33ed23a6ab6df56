"""Developer reference point: the Linear shapes of a denoise step on this library against torch's F.linear (hipBLASLt)
on the same tensors - a measurement of how far the shapes themselves allow one to go, not a product path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
from tools.op_bench import timeit, rnd
for B in (1, 4):
    shapes = [(1024 * B, 1280, 1280), (1024 * B, 1280, 3840), (1024 * B, 1280, 10240), (1024 * B, 5120, 1280),
              (4096 * B, 640, 640), (4096 * B, 640, 1920), (4096 * B, 640, 5120), (4096 * B, 2560, 640), (77 * B, 2048, 1280), (77 * B, 2048, 640)]
    tot_o = tot_v = 0.0
    for M, K, N in shapes:
        x, b = rnd(M, K), rnd(N)
        ncopy = max(1, min(32, int(600e6 // (N * K * 2))))
        ws = [rnd(N, K) * K ** -0.5 for _ in range(ncopy)]
        it = [0]
        def ours():
            it[0] += 1
            return ops.linear(x, ws[it[0] % ncopy], b)
        def vendor():
            it[0] += 1
            return torch.nn.functional.linear(x, ws[it[0] % ncopy], b)
        uo, uv = timeit(ours, iters=max(20, ncopy)), timeit(vendor, iters=max(20, ncopy))
        fl = 2.0 * M * K * N
        tot_o += uo; tot_v += uv
        print(f"B={B} M={M:6d} K={K:5d} N={N:5d}: ours {uo:7.1f} us {fl/uo/1e6:7.1f} TF/s | hipBLASLt {uv:7.1f} us {fl/uv/1e6:7.1f} TF/s | ratio {uv/uo:5.2f}", flush=True)
    print(f"B={B} sum: ours {tot_o:.0f} us, hipBLASLt {tot_v:.0f} us")
