"""Developer probe of attn32s_kernel (needs -DST_PROBE -DST_DEV_CONFIGS build): per-wave cycles of the two phases and their barrier waits."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import _C, ops
lib = _C.load()
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
    print(f"wave {w}: V work {int(p[w,0])/n:7.0f}  V wait+barrier {int(p[w,1])/n:7.0f}  M work {int(p[w,2])/n:7.0f}  M barrier {int(p[w,3])/n:7.0f}  cycles per trip, {n} trips")
