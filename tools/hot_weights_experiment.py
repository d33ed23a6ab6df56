"""Experiment: how much of the step time is HBM latency/bandwidth of cold weights?
All transformer layers / resnets of equal shape share one storage (numerics are meaningless, timing only)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import synth
from stabletriton_amd.optimization import optimize_model
from stabletriton_amd.pipeline import DenoiseLoop
from stabletriton_amd.scheduler import euler_discrete_tables
from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel
share = len(sys.argv) > 1 and sys.argv[1] == "share"
dev = torch.device("cuda:0"); dtype = torch.bfloat16
with torch.device("meta"):
    model = UNet2DConditionModel(SDXL_BASE)
model = model.to_empty(device=dev).to(dtype).eval().requires_grad_(False)
synth.fill_module_(model, 0)
if share:
    first = {}
    import re
    for name, p in model.named_parameters():
        key = (re.sub(r"\.\d+\.", ".N.", name), tuple(p.shape))
        if key in first:
            p.data = first[key].data
        else:
            first[key] = p
    print("distinct parameter storages:", len(first), "bytes", sum(p.numel() * 2 for p in first.values()) / 1e9, "GB")
gm = optimize_model(model, cuda_graph=False)
loop = DenoiseLoop(gm, 1, 128, dtype, dev, euler_discrete_tables(50), mode="loop")
x = synth.denoise_inputs(1, 128, 1234, device=dev)
loop.set_conditioning(x["encoder_hidden_states"].to(dtype), x["text_embeds"].to(dtype), x["time_ids"].to(dtype))
loop.set_noise(x["latent"])
with torch.no_grad():
    loop.run_steps(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop.run_steps(50); torch.cuda.synchronize()
    print("share" if share else "distinct", "ms/step", (time.perf_counter() - t0) / 50 * 1e3)
