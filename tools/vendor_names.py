"""Which hipBLASLt kernels torch's F.linear runs for the step's large Linear shapes (run under rocprofv3 --kernel-trace --stats:
the kernel names encode macro tile, K depth, workgroup and split).  Reference point for the eight-phase kernel."""
import sys
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
shapes = [(4096, 1280, 3840), (4096, 1280, 10240), (16384, 640, 5120), (16384, 2560, 640), (4096, 5120, 1280), (1024, 1280, 10240), (2048, 1280, 10240), (8192, 2560, 640)]
for M, K, N in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        y = F.linear(x, w, b)
    torch.cuda.synchronize()
    print("done", M, K, N, flush=True)
