"""Host hooks: drop the compiled UNet into the call sites that drive it.

Diffusers - reference call site implementations/Diffusers/load_sdxl_pipeline.py:17-46: an fp16 SDXL pipeline whose
`pipe.unet` is replaced by the compiled module with three `config` attributes re-attached.  The pipeline
(diffusers 0.21.2, third party) then calls, once per step and with classifier-free guidance (batch 2):

    unet(latent_model_input, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=None,
         added_cond_kwargs={"text_embeds": ..., "time_ids": ...}, return_dict=False)[0]

`DiffusersUNet` answers exactly that call.  With an fp16 module (the reference's own line: `.half().cuda()`) the HIP
kernels compute in fp16 and nothing is cast at this boundary; a bf16 or fp32 (strict parity) module takes the pipeline's
fp16 tensors through one cast in and one cast out.  The text-context K/V projections (140 GEMMs, step-invariant) are evaluated once per prompt - the cache is keyed
on the identity (a held reference, not the address) and version of `encoder_hidden_states` - and every step replays ONE captured hipGraph.

ComfyUI - the reference's hook is an empty file (implementations/ComfyUI/example.py), so the contract here is ComfyUI's
own `diffusion_model.forward(x, timesteps, context, y, control, transformer_options, **kw)`: `y` is the ready-made
2816-wide vector (pooled text embedding | Fourier features of the six size/crop ids), timesteps has one entry per row.
`ComfyUNet` wraps a module compiled from `UNetWithLabelVector` (same weights, `y` instead of added_cond_kwargs).

Both packages are duck-typed (neither is a dependency).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from .optimization import optimize_model
from .optimizers.graphs import make_dynamic_graphed_callable
from .unet import SDXL_BASE, UNet2DConditionModel, UNetSpec, UNetWithLabelVector, make_config


class _HoistedUNet(nn.Module):
    """Shared by both adapters: dtype casts at the boundary, per-prompt context cache, one captured graph per input shape."""

    def __init__(self, compiled, compute_dtype: torch.dtype, cuda_graph: bool = True):
        super().__init__()
        if not hasattr(compiled, "forward_with_context"):
            raise ValueError("expected a module from optimize_model(..., cuda_graph=False): the adapter captures its own graphs")
        self.compiled = compiled
        self.compute_dtype = compute_dtype
        self.cuda_graph = cuda_graph
        self._ctx: Dict[tuple, tuple] = {}        # ehs shape -> static context tensors (read by the captured graphs)
        # per shape: the tensor the cached context was built from, HELD (not its address: the caching allocator hands the
        # next prompt's tensor the address of a freed one, with _version 0 again), the version it had then, and a private
        # copy of its contents, to recognise the same prompt in a new tensor object.  Per shape: a caller that alternates
        # two context shapes (cond / uncond batches of different lengths) keeps both.
        self._ctx_src: Dict[tuple, torch.Tensor] = {}
        self._ctx_version: Dict[tuple, int] = {}
        self._ctx_copy: Dict[tuple, torch.Tensor] = {}
        self._steps: Dict[tuple, object] = {}     # ehs shape -> (graphed) step function
        self._new_prompt = False                  # set when a context was (re)projected: the fp8 plan measures its scales again
        self._last_t = None                       # fp8 plan only: the last call's (largest) timestep, to tell where a trajectory starts

    def refresh_weights(self) -> int:
        """Re-derive fused / folded weight buffers after an in-place weight update (also done at every new prompt).
        The hoisted text-context K/V were projected with the old weights: the next call recomputes them (and, with the fp8
        plan, measures the activation scales again)."""
        self._ctx_src.clear()
        self._ctx_copy.clear()
        return self.compiled.exec_context.refresh_derived(full=True)

    def _context_for(self, ehs: torch.Tensor) -> tuple:
        """Static K/V context buffers for this prompt.  Fast path: the very tensor object the cache was built from, at the
        version it had then.  A different object of the same shape (ComfyUI re-concatenates cond | uncond on every call) is
        compared BY CONTENT with the kept copy (one device compare + host sync, ~1 % of a step) and adopted when equal;
        anything else re-projects the context (one pass of 140 small GEMMs).  Under a stream capture of the caller's own
        the compare (a host sync) is skipped and the call counts as a new prompt."""
        shape = tuple(ehs.shape)
        if ehs is self._ctx_src.get(shape) and ehs._version == self._ctx_version.get(shape) and shape in self._ctx:
            return self._ctx[shape]
        copy = self._ctx_copy.get(shape)
        same = (copy is not None and shape in self._ctx and copy.dtype == ehs.dtype and copy.device == ehs.device
                and not torch.cuda.is_current_stream_capturing() and torch.equal(ehs, copy))
        if not same:
            self.compiled.exec_context.refresh_derived(full=True)
            with torch.no_grad():
                new = self.compiled.precompute_context(ehs.to(self.compute_dtype))
            old = self._ctx.get(shape)
            if old is None:
                self._ctx[shape] = tuple(t.clone() for t in new)
            else:
                for dst, src in zip(old, new):
                    dst.copy_(src)
            self._ctx_copy[shape] = ehs.detach().clone()
            self._new_prompt = True
        self._ctx_src[shape], self._ctx_version[shape] = ehs, ehs._version      # held: its address cannot be recycled under the cache
        return self._ctx[shape]

    def _step_fn(self, shape: tuple):
        fn = self._steps.get(shape)
        if fn is None:
            ctx = self._ctx[shape]                 # closed over: the graph reads these buffers in place

            def step(sample, timesteps, cond):
                return self.compiled.forward_with_context(sample, timesteps, ctx, cond)[0]

            fn = make_dynamic_graphed_callable(step) if self.cuda_graph else step
            self._steps[shape] = fn
        return fn

    def _fp8_restart(self, x, timesteps, shape, cond) -> None:
        """fp8 plan: the delayed activation scales restart from the first evaluation of every TRAJECTORY, as `DenoiseLoop` does
        (the hooks only restarted them when the prompt changed, so a new seed under the same prompt quantised its
        first step with the scales of the previous trajectory's last one).  A trajectory start is a new prompt or a timestep
        that is larger than the last call's (within a trajectory the timesteps fall); reading the timestep costs one host
        sync per call, paid only in this mode.  Under a stream capture of the caller's own nothing can be measured: a captured
        call replays whatever scales the last eager call left - evaluate the trajectory's first step eagerly before capturing."""
        if torch.cuda.is_current_stream_capturing():
            return
        t = float(timesteps.max())
        start = self._new_prompt or self._last_t is None or t > self._last_t
        self._last_t = t
        if start:
            from .optimization import recalibrate_fp8
            ctx = self._ctx[shape]
            run_once = lambda: self.compiled.forward_with_context(x, timesteps, ctx, cond)
            ectx = getattr(self.compiled, "exec_context", None)
            if ectx is not None and ectx.fp8 is None:
                with torch.no_grad():
                    run_once()          # the very first evaluation creates the scale slots; after it this trajectory starts like every later one
            recalibrate_fp8(self.compiled, run_once)

    def _run(self, sample, timesteps, ehs, cond):
        # (.to() is the identity when the caller already computes in this dtype: an fp16 pipeline over an fp16 module)
        io_dtype = sample.dtype
        dev = sample.device
        self._context_for(ehs)
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor(float(timesteps), dtype=torch.float32)
        timesteps = timesteps.to(device=dev, dtype=torch.float32)
        if timesteps.dim() > 1 or (timesteps.dim() == 1 and timesteps.numel() not in (1, sample.shape[0])):
            raise ValueError(f"timesteps of shape {tuple(timesteps.shape)} do not match batch {sample.shape[0]}")
        x = sample.to(self.compute_dtype)
        if getattr(self.compiled, "fp8_plan", False):
            self._fp8_restart(x, timesteps, tuple(ehs.shape), cond)
        self._new_prompt = False
        with torch.no_grad():
            out = self._step_fn(tuple(ehs.shape))(x, timesteps, cond)
        return out.to(io_dtype)


class DiffusersUNet(_HoistedUNet):
    """`pipe.unet` replacement (duck-typed `UNet2DConditionModel.forward` of diffusers 0.21.2)."""

    def __init__(self, compiled, spec: UNetSpec = SDXL_BASE, compute_dtype: torch.dtype = torch.bfloat16, cuda_graph: bool = True):
        super().__init__(compiled, compute_dtype, cuda_graph)
        self.config = make_config(spec)            # what the pipeline reads (load_sdxl_pipeline.py:29-34)

    @property
    def dtype(self):
        return self.compute_dtype

    def forward(self, sample, timestep, encoder_hidden_states=None, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None, return_dict: bool = False, **ignored):
        if encoder_hidden_states is None or added_cond_kwargs is None:
            raise ValueError("the SDXL UNet needs encoder_hidden_states and added_cond_kwargs{text_embeds, time_ids}")
        for name, val in (("class_labels", class_labels), ("timestep_cond", timestep_cond), ("attention_mask", attention_mask)):
            if val is not None:
                raise NotImplementedError(f"{name} is not part of the SDXL-base UNet path")
        if cross_attention_kwargs:
            # a pipeline with merged LoRA weights still passes {"scale": s}: scale 1 changes nothing once the weights are merged
            # (`refresh_weights()` after the merge); anything else would need attention processors the compiled graph has not
            extra = {k: v for k, v in cross_attention_kwargs.items() if not (k == "scale" and float(v) == 1.0)}
            if extra:
                raise NotImplementedError(f"cross_attention_kwargs {sorted(extra)} (attention processors / a LoRA scale other than 1) are not "
                                          "supported: merge the LoRA weights in place at the scale wanted and call refresh_weights()")
        cd = self.compute_dtype
        cond = {"text_embeds": added_cond_kwargs["text_embeds"].to(cd), "time_ids": added_cond_kwargs["time_ids"].to(cd)}
        out = self._run(sample, timestep, encoder_hidden_states, cond)
        if return_dict:
            from types import SimpleNamespace
            return SimpleNamespace(sample=out)
        return [out]                                # the pipeline takes [0] (reference unet_pt.py:542 returns a list)


def compile_unet_from_state_dict(state_dict, spec: UNetSpec = SDXL_BASE, dtype=None, device="cuda",
                                 cuda_graph: bool = True) -> DiffusersUNet:
    """Build the UNet, load a Diffusers-keyed state_dict (any float dtype), compile, wrap for the pipeline.
    `dtype` None = the state_dict's own dtype (fp16 for the reference's `variant="fp16"` checkpoint)."""
    if dtype is None:
        dtype = next(iter(state_dict.values())).dtype
    with torch.device("meta"):
        model = UNet2DConditionModel(spec)
    model = model.to_empty(device=device).to(dtype)
    model.load_state_dict({k: v.to(device=device, dtype=dtype) for k, v in state_dict.items()})
    compiled = optimize_model(model, cuda_graph=False)
    return DiffusersUNet(compiled, spec, dtype, cuda_graph)


def attach_to_diffusers(pipe, spec: UNetSpec = SDXL_BASE, dtype=None, cuda_graph: bool = True):
    """`pipe.unet = compiled UNet` (same weights; the counterpart of load_sdxl_pipeline.py:24-35), returns the pipeline.
    `dtype` None = the pipeline's own UNet dtype (fp16 at the reference call site): no casts at the boundary."""
    device = next(pipe.unet.parameters()).device
    pipe.unet = compile_unet_from_state_dict(pipe.unet.state_dict(), spec, dtype, device, cuda_graph)
    return pipe


class ComfyUNet(_HoistedUNet):
    """`diffusion_model` replacement with ComfyUI's calling convention."""

    def forward(self, x, timesteps=None, context=None, y=None, control=None, transformer_options: Optional[dict] = None, **kwargs):
        if timesteps is None or context is None or y is None:
            raise ValueError("ComfyUNet needs timesteps, context (text states) and y (pooled text | size/crop features)")
        if control is not None:
            raise NotImplementedError("ControlNet residuals (`control`) are not supported by the compiled UNet")
        patches = (transformer_options or {}).get("patches") or (transformer_options or {}).get("patches_replace")
        if patches:
            raise NotImplementedError("transformer_options patches are not supported by the compiled UNet")
        if y.shape[0] != x.shape[0] or context.shape[0] != x.shape[0]:
            raise ValueError("x, context and y must share the batch dimension")
        return self._run(x, timesteps, context, y.to(self.compute_dtype))


def compile_comfy_unet(unet: UNet2DConditionModel, cuda_graph: bool = True) -> ComfyUNet:
    """Compile the `y`-vector entry of a UNet (weights shared with `unet`)."""
    dtype = next(unet.parameters()).dtype
    compiled = optimize_model(UNetWithLabelVector(unet), cuda_graph=False)
    return ComfyUNet(compiled, dtype, cuda_graph)


def patch_comfy_model(model_patcher, unet: UNet2DConditionModel, cuda_graph: bool = True) -> ComfyUNet:
    """Replace `model_patcher.model.diffusion_model` (duck-typed ComfyUI ModelPatcher) with the compiled UNet."""
    adapter = compile_comfy_unet(unet, cuda_graph)
    model_patcher.model.diffusion_model = adapter
    return adapter
