"""Developer reference point: the 3x3 convolutions of a denoise step on this library against torch's conv2d (MIOpen),
channels-last bf16, same tensors - a measurement of what the shapes allow, not a product path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
from tools.op_bench import timeit, rnd
cl = torch.channels_last
torch.backends.cudnn.benchmark = True
for B in (1, 4):
    tot_o = tot_v = 0.0
    for Cin, H, Cout, cnt in ((320, 128, 320, 5), (320, 64, 640, 1), (640, 64, 640, 9), (640, 32, 1280, 1), (1280, 32, 1280, 13), (2560, 32, 1280, 3),
                              (1920, 32, 1280, 1), (1920, 64, 640, 1), (1280, 64, 640, 2), (960, 64, 640, 1), (960, 128, 320, 1), (640, 128, 320, 2)):
        x = rnd(B, Cin, H, H).contiguous(memory_format=cl)
        w = (rnd(Cout, Cin, 3, 3) * (Cin * 9) ** -0.5).contiguous(memory_format=cl)
        b = rnd(Cout)
        uo = timeit(lambda: ops.conv2d(x, w, b, 1, 1))
        uv = timeit(lambda: torch.nn.functional.conv2d(x, w, b, 1, 1))
        fl = 2.0 * B * H * H * Cout * Cin * 9
        tot_o += uo * cnt; tot_v += uv * cnt
        print(f"B={B} Cin={Cin:5d} HxW={H:3d} Cout={Cout:5d} x{cnt:2d}: ours {uo:7.1f} us {fl/uo/1e6:7.1f} TF/s | MIOpen {uv:7.1f} us {fl/uv/1e6:7.1f} TF/s | ratio {uv/uo:5.2f}", flush=True)
    print(f"B={B} weighted sum: ours {tot_o:.0f} us, MIOpen {tot_v:.0f} us")
