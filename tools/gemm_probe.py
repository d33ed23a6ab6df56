"""Developer diagnostic: phase cycle breakdown of the LDS-DMA GEMM (needs lib/probe/libst_probe.so,
built with -DST_PROBE).  Prints per-wave average cycles: compute+issue / vmcnt wait / barrier wait."""
import ctypes as C, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(root, "stabletriton_amd/lib/probe/libst_probe.so"))
lib.st_debug_set_probe.argtypes = [C.c_void_p]
p = C.c_void_p
lib.st_linear.argtypes = [p, p, p, p, p, p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, p, C.c_size_t, p, C.c_int, p, p]
dev = torch.device("cuda:0")
for (M, K, N) in [(1024, 1280, 1280), (1024, 5120, 1280), (4096, 640, 640), (1024, 1280, 3840), (77, 2048, 1280), (1, 1280, 1280)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    probe = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device=dev)
    lib.st_debug_set_probe(probe.data_ptr())
    for _ in range(3):
        lib.st_linear(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), M, N, K, K, N, 0, 0, 0, 1, None, 0, None, 0, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pr = probe.view(-1, 8).cpu()
    used = pr[pr[:, 7] > 0].double()
    nk = used[0, 7].item()
    span = (used[:, 6].max() - used[:, 5].min()).item()
    print(f"M={M} K={K} N={N}: waves={len(used)} nk={int(nk)} per-iter cycles: work={used[:,0].mean()/nk:.0f} vmwait={used[:,1].mean()/nk:.0f} "
          f"barrier={used[:,2].mean()/nk:.0f} | loop={used[:,3].mean():.0f} epilogue={used[:,4].mean():.0f} kernel-span={span:.0f} (100MHz ticks? see memtime)")
