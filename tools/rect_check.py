"""Developer check: the compiled UNet on RECTANGULAR latents (SDXL's usual aspect buckets: 1216 x 832 -> 152 x 104, 768 x 512 ->
96 x 64) - the tiny UNet against the oracle, SDXL bf16 against SDXL strict fp32 (same weights), eager and as a captured graph.
    python tools/rect_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import synth  # noqa: E402
from stabletriton_amd.optimization import optimize_model  # noqa: E402
from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel  # noqa: E402
from tests.test_unet_gpu import TINY, build  # noqa: E402
from oracle import unet_oracle as orc  # noqa: E402  (developer tool: the checker)

gpu = torch.device("cuda:0")


def inputs(batch, h, w, spec, seed=1234):
    x = synth.denoise_inputs(batch, 8, seed, cross_dim=spec.cross_dim, pooled_dim=spec.pooled_dim)
    x["latent"] = synth.normal("latent", (batch, 4, h, w), seed)
    x["time_ids"] = torch.tensor([[h * 8.0, w * 8.0, 0.0, 0.0, h * 8.0, w * 8.0]] * batch)
    return x


def run(gm, x, t, dtype):
    xg = {k: v.to(gpu, dtype) for k, v in x.items()}
    with torch.no_grad():
        return gm(xg["latent"], t.to(gpu), xg["encoder_hidden_states"], {"text_embeds": xg["text_embeds"], "time_ids": xg["time_ids"]})[0].float().cpu()


t = torch.tensor(321.0)
m, gm = build(TINY, torch.float32, gpu, graph=False)
sd = {k: v.float().cpu() for k, v in m.state_dict().items()}
for h, w in ((16, 24), (24, 8), (12, 20), (8, 40)):
    x = inputs(1, h, w, TINY)
    ref = orc.unet_forward(sd, x["latent"], t, x["encoder_hidden_states"], x["text_embeds"], x["time_ids"])
    out = run(gm, x, t, torch.float32)
    print(f"tiny fp32 {h}x{w}: max abs err {float((out - ref).abs().max()):.2e} (|ref| max {float(ref.abs().max()):.2f})", flush=True)
if "--sdxl" in sys.argv:
    m32, g32 = build(SDXL_BASE, torch.float32, gpu, graph=False)
    m16, g16 = build(SDXL_BASE, torch.bfloat16, gpu, graph=True)
    for h, w in ((96, 64), (152, 104), (104, 152), (128, 96)):
        x = inputs(1, h, w, SDXL_BASE)
        a = run(g32, x, t, torch.float32)
        b = run(g16, x, t, torch.bfloat16)
        b2 = run(g16, x, t, torch.bfloat16)
        print(f"sdxl {h}x{w}: strict rms {float(a.pow(2).mean().sqrt()):.3f} finite {bool(torch.isfinite(a).all())} | bf16 - strict rel rms "
              f"{float((b - a).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e} | replay identical {bool((b == b2).all())}", flush=True)
