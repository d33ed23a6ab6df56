"""Developer check: attention output while another stream keeps the CUs busy must equal the solo output bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, T, H) in ((1, 1024, 10), (1, 256, 20), (1, 4096, 10), (2, 1024, 20)):
    q, k, v = ((torch.randn(B, T, H * 64, device=dev)).bfloat16() for _ in range(3))
    solo = ops.attention(q, k, v, H, 0.125).clone()
    torch.cuda.synchronize()
    a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    side = torch.cuda.Stream(device=dev)
    bad = 0
    for it in range(20):
        with torch.cuda.stream(side):
            for _ in range(8):
                a2 = (a @ a).tanh_()
        out = ops.attention(q, k, v, H, 0.125)
        torch.cuda.synchronize()
        if not torch.equal(out, solo):
            bad += 1
            if bad == 1:
                d = (out.float() - solo.float()).abs().view(B, T, H, 64)
                rows = (d.amax(dim=(0, 2, 3)) > 0).nonzero().flatten()
                heads = (d.amax(dim=(0, 1, 3)) > 0).nonzero().flatten()
                print(f"  first mismatch: max |d| {d.max().item():.4g}, rows {rows[:8].tolist()}..{rows[-3:].tolist()} n={rows.numel()}, heads {heads.tolist()}, cols {(d.amax(dim=(0,1,2))>0).sum().item()}")
    print(f"B={B} T=S={T} H={H}: {bad}/20 runs differ from the solo run")
