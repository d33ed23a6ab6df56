"""Host hooks: drop the compiled UNet into a Diffusers SDXL pipeline.

Reference call site: implementations/Diffusers/load_sdxl_pipeline.py:24-35 -
build the minimal UNet, load the pipeline's state_dict, optimize_model, re-attach
the three `config` attributes the pipeline reads (lost by fx tracing), assign
`pipe.unet`.  The pipeline object is duck-typed (diffusers is not a dependency).
The reference's ComfyUI hook is an empty file (implementations/ComfyUI/example.py);
`patch_comfy_model` below is the minimal equivalent: it swaps the callable a
ComfyUI-style model patcher invokes for its diffusion model.
"""
from __future__ import annotations

import torch

from .optimization import optimize_model
from .unet import SDXL_BASE, UNet2DConditionModel, UNetSpec, make_config


def compile_unet_from_state_dict(state_dict, spec: UNetSpec = SDXL_BASE, dtype=torch.bfloat16, device="cuda",
                                 cuda_graph: bool = True):
    with torch.device("meta"):
        model = UNet2DConditionModel(spec)
    model = model.to_empty(device=device).to(dtype)
    model.load_state_dict({k: v.to(device=device, dtype=dtype) for k, v in state_dict.items()})
    compiled = optimize_model(model, cuda_graph=cuda_graph)
    compiled.config = make_config(spec)
    return compiled


def attach_to_diffusers(pipe, spec: UNetSpec = SDXL_BASE, dtype=torch.bfloat16, cuda_graph: bool = True):
    """`pipe.unet = compiled UNet` (same weights), returns the pipeline."""
    device = next(pipe.unet.parameters()).device
    pipe.unet = compile_unet_from_state_dict(pipe.unet.state_dict(), spec, dtype, device, cuda_graph)
    return pipe


class _ComfyAdapter(torch.nn.Module):
    """Callable with the Diffusers-style signature backing a ComfyUI-style wrapper
    `apply_model(x, t, c_crossattn=..., text_embeds=..., time_ids=...)`."""

    def __init__(self, compiled):
        super().__init__()
        self.compiled = compiled

    def forward(self, x, timesteps, context, text_embeds, time_ids, **ignored):
        t = timesteps.reshape(-1)[0] if timesteps.numel() > 1 else timesteps.reshape(())
        return self.compiled(x, t, context, {"text_embeds": text_embeds, "time_ids": time_ids})[0]


def patch_comfy_model(model_patcher, compiled) -> None:
    """Replace `model_patcher.model.diffusion_model` (duck-typed) with the compiled UNet."""
    model_patcher.model.diffusion_model = _ComfyAdapter(compiled)
