"""Developer probe of attn32i_kernel (needs a -DST_PROBE build, e.g. tools/build_one_variant.sh attprobe attention.hip -DST_PROBE;
run with ST_VARIANT=attprobe): per-wave cycles per trip of the QK phase, the PV phase, the lazy-maximum check and the DMA wait + barrier.
The stamps themselves drain the LDS queue (s_memtime + lgkmcnt(0)), so the phases read a little long."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import _C, ops
from tools.devlib import use_variant
lib = use_variant(os.environ.get("ST_VARIANT", "attprobe"))
dev = torch.device("cuda:0")
probe = torch.zeros(64, dtype=torch.int64, device=dev)
lib.st_debug_set_att_probe.argtypes = [ctypes.c_void_p]
lib.st_debug_set_att_probe(probe.data_ptr())
T, S, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 10)
q, k, v = (((torch.rand(1, n, H * 64, device=dev) * 2 - 1)).bfloat16() for n in (T, S, S))
for _ in range(3):
    ops.attention(q, k, v, H, 0.125)
torch.cuda.synchronize()
p = probe.cpu().view(8, 8)
for w in range(8):
    n = max(int(p[w, 4]), 1)
    print(f"wave {w}: QK {int(p[w,0])/n:7.0f}  PV {int(p[w,1])/n:7.0f}  check {int(p[w,2])/n:7.0f}  wait+barrier {int(p[w,3])/n:7.0f}  cycles per trip, {n} trips")
