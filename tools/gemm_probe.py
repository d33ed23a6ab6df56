"""Developer diagnostic: phase breakdown of the LDS-DMA GEMM (needs `tools/build_variant.sh probe -DST_PROBE`).
Prints per-wave averages per K trip in s_memtime ticks (shader-clock cycles on gfx950): issue+MFMA / vmcnt wait / barrier wait."""
import ctypes as C, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(root, "stabletriton_amd/lib/probe/libstabletriton_amd.so"))
lib.st_debug_set_probe.argtypes = [C.c_void_p]
p = C.c_void_p
lib.st_linear.argtypes = [p, p, p, p, p, p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, p, C.c_size_t, p, C.c_int, p, p]
dev = torch.device("cuda:0")
SHAPES = [(1024, 1280, 1280, 0), (1024, 1280, 5120, 1), (1024, 5120, 1280, 0), (4096, 640, 640, 0), (4096, 640, 2560, 1), (1024, 1280, 3840, 0), (77, 2048, 1280, 0)]
for (M, K, N, geglu) in SHAPES:
    rows = 2 * N if geglu else N
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(rows, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(rows, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    probe = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
    lib.st_debug_set_probe(probe.data_ptr())
    for _ in range(3):
        rc = lib.st_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), M, N, K, K, N, 0, 0, 1 | (4 if geglu else 0), 1, None, 0, None, 0, None,
                           torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    pr = probe.view(-1, 8).cpu()
    used = pr[pr[:, 7] > 0].double()
    nk = used[0, 7].item()
    print(f"M={M} K={K} N={N} geglu={geglu}: waves={len(used)} trips={int(nk)} per-trip cycles: work={used[:,0].mean()/nk:.0f} vmwait={used[:,1].mean()/nk:.0f} "
          f"barrier={used[:,2].mean()/nk:.0f} | loop={used[:,3].mean():.0f} epilogue={used[:,4].mean():.0f} (loads+math {used[:,5].mean():.0f}, stores {used[:,6].mean():.0f})")
