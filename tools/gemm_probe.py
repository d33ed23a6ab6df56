"""Developer diagnostic: phase breakdown of the LDS-DMA GEMM per wave, in shader-clock cycles (s_memtime).
Needs `tools/build_one_variant.sh probe gemm_api.hip gemm_dense_bf16.hip -DST_PROBE -DST_DEV_CONFIGS`.
usage: gemm_probe.py M K N cfg [cfg ...]      (cfg = CFG_* id of gemm_core.h, -1 = product dispatch)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.devlib import use_variant
lib = use_variant(os.environ.get("ST_VARIANT", "probe"))
from stabletriton_amd import ops
lib.st_debug_set_probe.argtypes = [C.c_void_p]
lib.st_debug_force_gemm.argtypes, lib.st_debug_force_gemm.restype = [C.c_int, C.c_int], None
dev = torch.device("cuda:0")
M, K, N = (int(v) for v in sys.argv[1:4])
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
for cfg in [int(c) for c in sys.argv[4:]] or [-1]:
    lib.st_debug_force_gemm(cfg, 1 if cfg >= 0 else -1)
    probe = torch.zeros(8192 * 8 * 12, dtype=torch.int64, device=dev)
    lib.st_debug_set_probe(probe.data_ptr())
    for _ in range(3):
        probe.zero_()
        ops.linear(x, w, b)
    torch.cuda.synchronize()
    pr = probe.view(-1, 12).cpu()
    used = pr[pr[:, 7] > 0].double()
    if len(used) == 0:
        print(f"cfg {cfg}: kernel writes no probe (not gemm_dma_kernel)"); continue
    nk = used[0, 7].item()
    if used[0, 11].item() == 1:         # gemm8p_kernel: epilogue broken down
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.linear(x, w, b)
        e1.record(); torch.cuda.synchronize()
        m = lambda c: used[:, c].mean().item()
        print(f"M={M} K={K} N={N} cfg={cfg} (eight-phase): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us eager; waves={len(used)} trips={int(nk)} cycles: prologue={m(8):.0f} loop={m(3):.0f} "
              f"({m(3) / nk:.0f} per trip) epilogue={m(4):.0f} = operands+first barrier {m(0):.0f} + park chunk 0 + barrier {m(1):.0f} + process chunk 0 {m(2):.0f} + closing barrier {m(5):.0f} "
              f"+ other chunks {m(6):.0f} | wall (100 MHz ticks -> us): first entry -> last entry {(used[:,9].max()-used[:,9].min())/100:.2f}, "
              f"first entry -> last exit {(used[:,10].max()-used[:,9].min())/100:.2f}, mean block life {(used[:,10]-used[:,9]).mean()/100:.2f}")
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(x, w, b)
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} K={K} N={N} cfg={cfg}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us eager; waves={len(used)} trips={int(nk)} per-trip cycles: work={used[:,0].mean()/nk:.0f} vmwait={used[:,1].mean()/nk:.0f} "
          f"barrier={used[:,2].mean()/nk:.0f} | prologue={used[:,8].mean():.0f} loop={used[:,3].mean():.0f} (max {used[:,3].max():.0f}) epilogue={used[:,4].mean():.0f} "
          f"(loads+park+barrier {used[:,5].mean():.0f}, vectors+stores {used[:,6].mean():.0f}) | wall (100 MHz ticks -> us): first entry -> last entry {(used[:,9].max()-used[:,9].min())/100:.2f}, "
          f"first entry -> last exit {(used[:,10].max()-used[:,9].min())/100:.2f}, mean block life {(used[:,10]-used[:,9]).mean()/100:.2f}")
