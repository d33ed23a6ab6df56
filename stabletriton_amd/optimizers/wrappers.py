"""fx leaf functions: the operator boundary of the compiled graph.

Names and argument meaning follow the reference's `torch.fx.wrap`-registered
wrappers so a graph rewritten by either project calls the same symbols
(SURVEY.md 8b): attention_wrapper (optimizers/replace_attention.py:60-71),
geglu_triton (replace_geglu.py:23-30), group_norm_wrapper
(replace_groupnorm.py:18-21), layer_norm_wrapper (replace_layernorm.py:17-27),
linear_wrapper / linear_wrapper_functional (replace_linear.py:20-37),
timestep_wrapper (replace_timesteps.py:33-40).  Here each one launches a
hand-written gfx950 kernel through the C ABI (stabletriton_amd/ops.py).
Unlike the reference, wrappers never mutate the module's parameters
(replace_layernorm.py:19-22 / replace_linear.py:28-32 down-cast them in place).

conv2d_wrapper and the *_fused wrappers are additions: the reference leaves
convolutions to cuDNN and the epilogues unfused (optimizations.txt:5,13-66).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import ops


def attention_wrapper(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, output: Optional[torch.Tensor],
                      sm_scale: float, num_heads: int, head_dim: int) -> torch.Tensor:
    """q,k,v in (B, T|S, H*D) projection layout; `output` is accepted and ignored
    exactly as in the reference."""
    if q.shape[-1] != num_heads * head_dim:
        raise ops.BackendError(f"attention_wrapper: C={q.shape[-1]} != num_heads*head_dim={num_heads * head_dim}")
    return ops.attention(q, k, v, num_heads, sm_scale)


def geglu_triton(state: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    return ops.geglu(state, gate)


def group_norm_wrapper(v: torch.Tensor, groupnorm: nn.GroupNorm, activation: bool) -> torch.Tensor:
    return ops.group_norm(v, groupnorm.num_groups, groupnorm.weight, groupnorm.bias, groupnorm.eps, activation)


def layer_norm_wrapper(v: torch.Tensor, layernorm: nn.LayerNorm) -> torch.Tensor:
    return ops.layer_norm(v, layernorm.weight, layernorm.bias, layernorm.eps)


def linear_wrapper_functional(v: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                              activation: bool, emit_stats: bool = False):
    return ops.linear(v, weight, bias, silu=activation, emit_stats=emit_stats)


def linear_wrapper(v: torch.Tensor, linear: nn.Linear, activation: bool, emit_stats: bool = False):
    """`emit_stats` (addition): also return the LayerNorm partials of the output rows, for a
    following `ln_linear_wrapper`; the result is then the pair (out, stats)."""
    return linear_wrapper_functional(v, linear.weight, linear.bias, activation, emit_stats)


def timestep_wrapper(x: torch.Tensor):
    """The reference's leaf of the same name (optimizers/replace_timesteps.py:33-37 -> kernels/timestep.py:13-45):
    x is the timestep tensor already broadcast to (..., half); returns (sin, cos) of x * f_j elementwise,
    f_j = exp(-ln(1e4) * j / half) along the last dimension."""
    return ops.timestep_sincos(x)


# ---- additions -------------------------------------------------------------------------------
def timestep_embedding_wrapper(t: torch.Tensor, dim: int) -> torch.Tensor:
    """cos|sin features of a 1-D fp32 timestep tensor, (len(t), dim) fp32: the whole sinusoidal sub-graph of
    unet_pt.py:22-36 as one launch (the reference's fuse_timesteps pass matches 0 sites of the real UNet)."""
    return ops.timestep_features(t, dim, torch.float32)


def conv2d_wrapper(v: torch.Tensor, conv: nn.Conv2d, upsample2x: bool = False,
                   rowbias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, emit_colstats: bool = False):
    """`emit_colstats` (addition): also return the GroupNorm partials of the output channels, for a following
    `group_norm_stats_wrapper`; the result is then the pair (out, stats)."""
    s, p = conv.stride, conv.padding
    if s[0] != s[1] or p[0] != p[1] or conv.dilation != (1, 1) or conv.groups != 1 or isinstance(p, str):
        raise ops.BackendError("conv2d_wrapper: only square stride/padding, dilation 1, groups 1")
    return ops.conv2d(v, conv.weight, conv.bias, s[0], p[0], upsample2x=upsample2x, rowbias=rowbias, residual=residual,
                      emit_colstats=emit_colstats)


def group_norm_stats_wrapper(v, stats, groupnorm: nn.GroupNorm, activation: bool) -> torch.Tensor:
    """group_norm_wrapper whose statistics come from the launch(es) that produced `v` (`stats`: one ColStats, or two
    for a channel concatenation): finalize + apply, no statistics pass over v.  `v` may be the pair of tensors of a
    channel concatenation that was never written (optimizers/fuse_skip_cat.py)."""
    if isinstance(v, (tuple, list)):
        return ops.group_norm_from_stats_cat(v[0], v[1], tuple(stats), groupnorm.num_groups, groupnorm.weight, groupnorm.bias,
                                             groupnorm.eps, activation)
    return ops.group_norm_from_stats(v, tuple(stats), groupnorm.num_groups, groupnorm.weight, groupnorm.bias, groupnorm.eps, activation)


def conv2d_cat_wrapper(a: torch.Tensor, b: torch.Tensor, conv: nn.Conv2d, residual: Optional[torch.Tensor] = None,
                       emit_colstats: bool = False):
    """conv2d_wrapper(torch.cat([a, b], 1), conv) for a 1x1 / stride-1 / unpadded conv, without the concatenated tensor."""
    return ops.conv2d_cat(a, b, conv.weight, conv.bias, residual=residual, emit_colstats=emit_colstats)


def linear_geglu_wrapper(v: torch.Tensor, linear: nn.Linear) -> torch.Tensor:
    """GEGLU projection with the gate applied in the GEMM epilogue (unet_pt.py:155-158)."""
    return ops.linear(v, linear.weight, linear.bias, geglu=True)


def linear_residual_wrapper(v: torch.Tensor, linear: nn.Linear, residual: torch.Tensor, emit_stats: bool = False,
                            emit_colstats: bool = False):
    return ops.linear(v, linear.weight, linear.bias, residual=residual, emit_stats=emit_stats, emit_colstats=emit_colstats)


for _name in ("attention_wrapper", "geglu_triton", "group_norm_wrapper", "layer_norm_wrapper", "linear_wrapper",
              "linear_wrapper_functional", "timestep_wrapper", "timestep_embedding_wrapper", "conv2d_wrapper", "linear_geglu_wrapper",
              "linear_residual_wrapper", "group_norm_stats_wrapper", "conv2d_cat_wrapper"):
    torch.fx.wrap(_name)


# ---- fused projections sharing one input (q|k|v of self-attention, k|v of cross-attention) ----
def _cat_weight(linears):
    """Row-concatenated weight (and bias) of several nn.Linear with the same in_features.  The buffer is owned
    by the current execution context (the compiled module) and is refreshed IN PLACE when a source parameter
    changes, so weight swaps / LoRA merges stay visible and addresses captured in hipGraphs stay valid."""
    sources = [l.weight for l in linears] + [l.bias for l in linears if l.bias is not None]

    def compute():
        w = torch.cat([l.weight.detach() for l in linears], dim=0).contiguous()
        if all(l.bias is None for l in linears):
            b = None
        else:
            b = torch.cat([l.bias.detach() if l.bias is not None else
                           torch.zeros(l.out_features, dtype=l.weight.dtype, device=l.weight.device) for l in linears])
        return (w, b)

    key = ("cat",) + tuple(id(l) for l in linears)
    return ops.current_context(linears[0].weight.device).derived_weights(key, sources, compute).value


def linear_cat_wrapper(v: torch.Tensor, linears) -> torch.Tensor:
    """[lin_0(v) | lin_1(v) | ...] along the last dimension, as ONE GEMM."""
    w, b = _cat_weight(tuple(linears))
    return ops.linear(v, w, b)


torch.fx.wrap("linear_cat_wrapper")


# ---- LayerNorm folded into the projection(s) that consume it ---------------------------------
def _ln_fold(layernorm: nn.LayerNorm, linears):
    sources = [layernorm.weight, layernorm.bias] + [l.weight for l in linears] + [l.bias for l in linears if l.bias is not None]

    @torch.no_grad()
    def compute():
        w = torch.cat([l.weight.detach() for l in linears], dim=0) if len(linears) > 1 else linears[0].weight.detach()
        if all(l.bias is None for l in linears):
            b = None
        else:
            b = torch.cat([l.bias.detach().float() if l.bias is not None else
                           torch.zeros(l.out_features, dtype=torch.float32, device=w.device) for l in linears])
        return ops.fold_layer_norm(layernorm.weight.detach(), layernorm.bias.detach(), w, b)

    key = ("ln", id(layernorm)) + tuple(id(l) for l in linears)
    return ops.current_context(layernorm.weight.device).derived_weights(key, sources, compute).value


def ln_linear_wrapper(v: torch.Tensor, stats, layernorm: nn.LayerNorm, linears, geglu: bool = False) -> torch.Tensor:
    """linear_i(layernorm(v)) for every linear in `linears`, concatenated on the last dimension
    (or the GEGLU of the single projection), as ONE GEMM: the LayerNorm never runs as its own
    kernel.  `stats` are the row partials the GEMM that produced `v` emitted (emit_stats=True)."""
    wf, c, d = _ln_fold(layernorm, tuple(linears))
    # (three projections = q|k|v of self-attention: in strict mode the K / V columns meet the attention kernel as split images)
    return ops.ln_linear(v, stats, wf, c, d, layernorm.eps, geglu=geglu, emit_split=len(linears) == 3)


torch.fx.wrap("ln_linear_wrapper")


def ln_linear_attention_wrapper(v: torch.Tensor, stats, layernorm: nn.LayerNorm, linear: nn.Linear, k: torch.Tensor, v_ctx: torch.Tensor,
                                sm_scale: float, num_heads: int, head_dim: int) -> torch.Tensor:
    """attention_wrapper(ln_linear_wrapper(v, stats, layernorm, (linear,)), k, v_ctx, None, sm_scale, num_heads, head_dim)
    - the cross-attention query path of a transformer block (norm2 -> attn2.to_q -> attention over the hoisted text
    context) - as ONE launch when the shapes allow it (bf16, head_dim 64, context < 256 tokens, 128 | tokens) and the
    launch is at most one round of its 128 x 64 tiles, else as those two calls.  Both routes give the same bits."""
    wf, c, d = _ln_fold(layernorm, (linear,))
    if head_dim == 64 and ops.xattn_fusable(v, k, num_heads) and ops.xattn_fusion_pays(v, num_heads):
        return ops.ln_linear_xattn(v, stats, wf, c, d, layernorm.eps, k, v_ctx, num_heads, sm_scale)
    q = ops.ln_linear(v, stats, wf, c, d, layernorm.eps)
    return ops.attention(q, k, v_ctx, num_heads, sm_scale)


torch.fx.wrap("ln_linear_attention_wrapper")
