"""Developer diagnostic: per-tile phase cycles of the bf16 attention kernel (lib/probe build)."""
import ctypes as C, os
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(root, "stabletriton_amd/lib/probe/libstabletriton_amd.so"))
p = C.c_void_p
lib.st_debug_set_att_probe.argtypes = [p]
lib.st_attention.argtypes = [p, p, p, p] + [C.c_int] * 5 + [C.c_long] * 4 + [C.c_float, C.c_int, p]
dev = torch.device("cuda:0")
for (T, S, H) in [(4096, 4096, 10), (1024, 1024, 20), (1024, 77, 20)]:
    q = torch.randn(1, T, H * 64, device=dev).bfloat16(); k = torch.randn(1, S, H * 64, device=dev).bfloat16(); v = torch.randn_like(k)
    o = torch.empty_like(q)
    probe = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
    lib.st_debug_set_att_probe(probe.data_ptr())
    for _ in range(3):
        lib.st_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 1, T, S, H, 64, H * 64, H * 64, H * 64, H * 64, 0.125, 1,
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pr = probe.view(-1, 8).cpu().double()
    u = pr[pr[:, 4] > 0]
    n = u[0, 4].item()
    print(f"T={T} S={S} H={H}: waves={len(u)} tiles={int(n)} per-tile cycles: wait+barrier+dma+QKnext={u[:,0].mean()/n:.0f} softmax={u[:,1].mean()/n:.0f} "
          f"PV={u[:,2].mean()/n:.0f} (unused)={u[:,3].mean()/n:.0f}")
