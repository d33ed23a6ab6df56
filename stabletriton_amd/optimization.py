"""Compile entry point: `model = optimize_model(model, cuda_graph)`.

Same surface as reference src/stabletriton/optimization.py:10-38 (README.md:5
calls it `compile`): trace the eager module with torch.fx, run the rewrite
passes in a fixed order (fused variants claim their nodes first), optionally
wrap `forward` in the shape-keyed hipGraph cache.  The returned GraphModule is
callable with the original forward signature; the caller re-attaches `.config`
(implementations/Diffusers/load_sdxl_pipeline.py:29-34) or uses
`stabletriton_amd.hooks`.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import fx, nn

from . import _C, ops
from .optimizers import (dedupe_pure_calls, fuse_token_residual, fuse_attention, fuse_geglu, fuse_geglu_into_linear, fuse_groupnorm_stats, fuse_skip_cat, fuse_layernorm_into_linear, fuse_query_projection_into_attention, fuse_residual_adds,
                         fuse_shared_input_linears,
                         fuse_temb_add, fuse_timesteps, split_context, split_region, keep_channels_last, make_dynamic_graphed_callable, plan_fp8, remove_dropout,
                         replace_conv, replace_group_norm, replace_group_norm_activation, replace_layer_norm,
                         replace_linear, replace_linear_activ)


def replace_backend(gm: fx.GraphModule, fuse: bool = True, fp8: bool = False, xattn_fusion: bool = True,
                    gn_stats: bool = True) -> fx.GraphModule:
    """Pass pipeline.  The first eight passes and their order are the reference's
    (optimization.py:10-22); replace_linear is enabled (the MFMA GEMM is the
    product here), replace_conv / epilogue fusions / layout are additions.
    `xattn_fusion` / `gn_stats` switch two of the added fusions off (A/B measurements)."""
    stats: Dict[str, int] = {}
    stats["dropout"] = remove_dropout(gm)
    if fuse:
        stats["deduped_activations"] = dedupe_pure_calls(gm)
    stats["attention"] = fuse_attention(gm)
    stats["geglu"] = fuse_geglu(gm)
    stats["linear_silu"] = replace_linear_activ(gm, nn.SiLU())
    stats["group_norm_silu"] = replace_group_norm_activation(gm, nn.SiLU())
    stats["group_norm"] = replace_group_norm(gm)
    stats["layer_norm"] = replace_layer_norm(gm)
    stats["linear"] = replace_linear(gm)
    stats["timesteps"] = fuse_timesteps(gm)
    stats["conv"] = replace_conv(gm)
    if fuse:
        stats["geglu_in_gemm"] = fuse_geglu_into_linear(gm)
        stats["temb_rowbias"] = fuse_temb_add(gm)
        stats["residual_adds"] = fuse_residual_adds(gm)
        stats["token_residuals"] = fuse_token_residual(gm)
        stats["shared_input_gemms"] = fuse_shared_input_linears(gm)
        stats["layer_norm_in_gemm"] = fuse_layernorm_into_linear(gm)
        stats["query_projection_in_attention"] = fuse_query_projection_into_attention(gm) if xattn_fusion else 0
        stats["group_norm_stats"] = fuse_groupnorm_stats(gm) if gn_stats else 0
        stats["skip_cats_removed"] = fuse_skip_cat(gm) if gn_stats else 0      # (the decoder's torch.cat: both readers take the two halves)
        if fp8:      # the three big projections of every transformer block on the fp8 matrix pipe, fed by e4m3 copies their
            stats["fp8_plan"] = plan_fp8(gm)      # producers' epilogues leave (no quantisation launches): optimizers/plan_fp8.py
    stats["channels_last_views"] = keep_channels_last(gm)
    gm.graph.eliminate_dead_code()
    gm.graph.lint()
    gm.recompile()
    gm.rewrite_stats = stats
    return gm


def run_compiler(gm: fx.GraphModule) -> fx.GraphModule:
    return replace_backend(gm)


def optimize_model(model: nn.Module, cuda_graph: bool = True, fuse: bool = True, fp8: bool = False) -> fx.GraphModule:
    """`fp8=True` (addition, BASELINE config #5): the q|k|v, GEGLU and feed-forward output projections of a bf16 model's
    transformer blocks run with OCP e4m3 operands on the fp8 matrix pipe, fed by e4m3 copies their producers' epilogues
    leave under delayed per-tensor scales (optimizers/plan_fp8.py); everything else is unchanged."""
    # same preconditions as the reference (optimization.py:29-33), for ROCm
    assert torch.cuda.is_available(), "a ROCm GPU is required to use stabletriton_amd"
    major, _ = torch.cuda.get_device_capability()
    if major < 9:
        raise RuntimeError("a CDNA GPU (gfx9xx; built and tuned for gfx950 / MI355X) is required")
    p0 = next(model.parameters())
    assert p0.device.type == "cuda", "Model must be on GPU"
    # fp16 is what the reference's call site passes (load_sdxl_pipeline.py:17-28: `.half().cuda()`); bf16 has the same
    # matrix-pipe rate and fp32's exponent range; fp32 is the strict parity mode
    if p0.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise RuntimeError(f"model dtype {p0.dtype} not supported: use float16 / bfloat16 (fast) or float32 (strict parity)")
    _C.load()                                  # fail now, loudly, if the HIP library is missing
    model = model.eval().to(memory_format=torch.channels_last)      # conv weights -> (Cout,R,S,Cin) strides
    if fp8 and p0.dtype != torch.bfloat16:
        raise RuntimeError("fp8 projections need a bfloat16 model")
    gm = replace_backend(fx.symbolic_trace(model), fuse=fuse, fp8=fp8)
    # the compiled module owns its mutable host state (split-K workspace, next-weights plan, derived weight buffers):
    # two compiled modules, or two streams each driving their own, never share any (ops.ExecContext)
    gm.exec_context = ops.ExecContext()
    gm.fp8_plan = bool(fp8 and fuse)
    # the host's tables of the sinusoidal timestep features (ops._timestep_table): built now, eagerly, so that no first call
    # inside somebody's stream capture has to compute them the other way
    from .optimizers.wrappers import timestep_embedding_wrapper
    for n in gm.graph.nodes:
        if n.op == "call_function" and n.target is timestep_embedding_wrapper and isinstance(n.args[1], int):
            ops._timestep_table(p0.device, n.args[1])
    if not (fuse and _install_context_split(gm)):
        plain = gm.forward

        def forward(*args, **kwargs):
            return _run_step(gm, lambda: plain(*args, **kwargs))

        gm.forward = forward
    if cuda_graph:
        gm.forward = make_dynamic_graphed_callable(gm.forward, before_replay=gm.exec_context.refresh_derived)
    return gm


def _run_step(gm: fx.GraphModule, core):
    """One UNet evaluation inside the module's step scope.  With the fp8 plan: the launch that turns the previous step's
    maxima into this step's scales goes first; the very first evaluation runs twice (its first pass only measures)."""
    ectx = gm.exec_context
    with ectx.step():
        if not getattr(gm, "fp8_plan", False):
            return core()
        if ectx.fp8 is None or not ectx.fp8.calibrated:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("fp8 plan: run one eager step before capturing (DenoiseLoop.capture and the graph cache do)")
            out = core()                             # measuring pass: e4m3 copies under the initial scales, true maxima recorded
            if ectx.plan is not None:
                ectx.plan.reset()
            if ectx.fp8 is None:                     # nothing in this module qualified for the plan
                gm.fp8_plan = False
                return out
            ectx.fp8.calibrated = True
    with ectx.step():
        ectx.fp8.update()
        return core()


FP8_CALIBRATION_PASSES = 2      # measuring evaluations after a reset: the first runs under the initial scale (values beyond 28 clip,
                                # so what it records downstream of a clipped tensor is too small), the second under scales from the first


def recalibrate_fp8(gm, run_once) -> bool:
    """Start the delayed scales of `gm`'s fp8 plan over and measure them on the evaluation `run_once()` performs (eagerly,
    outside any capture; its result is discarded): after this the scales are a function of that evaluation's inputs alone,
    so two identical trajectories give identical results whatever ran before.  Returns False when there is nothing to do."""
    ectx = getattr(gm, "exec_context", None)
    if ectx is None or not getattr(gm, "fp8_plan", False) or ectx.fp8 is None:
        return False
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("fp8 plan: the scales cannot be re-measured inside a graph capture")
    ectx.fp8.reset()
    with torch.no_grad():
        for _ in range(FP8_CALIBRATION_PASSES):
            ectx.fp8.calibrated = False            # _run_step: one measuring pass, the scale update, one pass under the new scales
            run_once()
    return True


def _install_context_split(gm: fx.GraphModule) -> bool:
    """Hoist the text-context projections (step-invariant) out of the per-step graph.

    `gm(sample, timesteps, encoder_hidden_states, added_cond_kwargs)` keeps working and stays
    stateless (it evaluates the context part on every call).  Loops that reuse one prompt call
    `ctx = gm.precompute_context(ehs)` once and `gm.forward_with_context(sample, t, ctx, cond)`
    per step (stabletriton_amd/pipeline.py)."""
    names = [n.target for n in gm.graph.nodes if n.op == "placeholder"]
    if "encoder_hidden_states" not in names:
        return False
    context_module = split_context(gm, "encoder_hidden_states")
    if context_module is None:
        return False
    if getattr(gm, "exec_context", None) is None:
        gm.exec_context = ops.ExecContext()
    ectx = gm.exec_context
    gm.rewrite_stats["context_outputs"] = len(
        [n for n in context_module.graph.nodes if n.op == "output"][0].args[0])
    gm.context_module = context_module
    # The time path (sinusoidal features -> TimestepEmbedding MLPs -> the batched resnet time_emb_proj)
    # depends only on the timestep and the added conditioning: a loop evaluates it once per schedule
    # entry and feeds the per-step row back in (pipeline.DenoiseLoop).
    time_module = None
    cond_arg = "added_cond_kwargs" if "added_cond_kwargs" in names else ("y" if "y" in names else None)
    if "timesteps" in names and cond_arg is not None:
        time_module = split_region(gm, ("timesteps", cond_arg), "time_cache", meta_sources=("sample",),
                                   stop_at_slices=True, class_name="TimeModule")
    core = gm.forward     # generated: (sample, timesteps, ehs, context_cache, added_cond_kwargs[, time_cache], **kw)

    def precompute_context(encoder_hidden_states):
        with ectx:
            return context_module(encoder_hidden_states)

    if time_module is None:
        def forward_with_context(sample, timesteps, context_cache, added_cond_kwargs, **kwargs):
            return _run_step(gm, lambda: core(sample, timesteps, None, context_cache, added_cond_kwargs, **kwargs))
    else:
        gm.time_module = time_module
        gm.rewrite_stats["time_outputs"] = len([n for n in time_module.graph.nodes if n.op == "output"][0].args[0])

        def precompute_time(sample, timesteps, added_cond_kwargs):
            """Time-path tensors of one schedule entry (`sample` is read for its batch size and dtype only)."""
            with ectx:
                return time_module(sample, timesteps, added_cond_kwargs)

        def forward_with_context(sample, timesteps, context_cache, added_cond_kwargs, time_cache=None, **kwargs):
            if time_cache is None:
                time_cache = precompute_time(sample, timesteps, added_cond_kwargs)
            # one pass over the same GEMMs in the same order: each launch warms the next one's weights
            return _run_step(gm, lambda: core(sample, timesteps, None, context_cache, added_cond_kwargs, time_cache, **kwargs))

        gm.precompute_time = precompute_time

    def forward(sample, timesteps, encoder_hidden_states, added_cond_kwargs, **kwargs):
        ctx = precompute_context(encoder_hidden_states)      # (outside the per-step weight plan: a loop evaluates it once)
        return forward_with_context(sample, timesteps, ctx, added_cond_kwargs, **kwargs)

    gm.precompute_context = precompute_context
    gm.forward_with_context = forward_with_context
    gm.forward = forward
    return True


compile = optimize_model
