"""Does the block-scaled form of the fp8 matrix instruction (E8M0 scale per 32 k: MX) buy accuracy over ONE delayed scale per
activation tensor?  CPU experiment, e4m3 arithmetic emulated with torch.float8_e4m3fn: a LayerNorm-fed projection of the SDXL
shape (1024 x 1280 -> 1280) on (a) unit-normal activations, (b) activations with outlier channels (a few columns 30x the
rest: the residual stream in front of a LayerNorm fold), (c) a heavy-tailed tensor.  Weights per output channel in every case.
usage: python tools/mx_scale_experiment.py"""
import torch

torch.manual_seed(0)
M, K, N = 1024, 1280, 1280
E4 = torch.float8_e4m3fn


def q_tensor(x):
    s = x.abs().max() / 448.0 * 2.0          # the product's delayed scale: previous maximum x margin 2
    return (x / s).clamp(-448, 448).to(E4).float() * s


def q_block(x, blk=32):
    xb = x.view(x.shape[0], -1, blk)
    s = torch.exp2(torch.ceil(torch.log2(xb.abs().amax(-1, keepdim=True).clamp_min(1e-30) / 448.0)))      # E8M0: powers of two
    return ((xb / s).clamp(-448, 448).to(E4).float() * s).view_as(x)


def q_weight(w):
    s = w.abs().amax(1, keepdim=True) / 448.0
    return (w / s).to(E4).float() * s


def rel_rms(a, b):
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


w = torch.randn(N, K) * K ** -0.5
wq = q_weight(w)
cases = {"unit normal": torch.randn(M, K)}
x = torch.randn(M, K); x[:, torch.randperm(K)[:8]] *= 30.0
cases["8 outlier channels x30"] = x
cases["heavy tails (normal^3)"] = torch.randn(M, K) ** 3
for name, x in cases.items():
    ref = x @ w.T
    print(f"{name:26s}: per-tensor scale {rel_rms(q_tensor(x) @ wq.T, ref):.4f}   E8M0 per 32 k {rel_rms(q_block(x) @ wq.T, ref):.4f}   "
          f"(weights only quantised {rel_rms(x @ wq.T, ref):.4f})")
