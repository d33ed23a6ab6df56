"""fp8 projection mode (addition; SURVEY.md 8f-4, BASELINE config #5).

The reference's seed for this is its unused fused-projection kernel, which stores the projection weights as fp8 and
up-converts them before the product (kernels/attention_proj.py:36-39, 105-155).  Here the projections of the
transformer blocks (to_q / to_k / to_v, to_out, the GEGLU projection, the feed-forward output, proj_in / proj_out)
run on the fp8 matrix pipe with both operands in OCP e4m3: weights are quantised once per output channel, activations
row by row in the pass that produces the GEMM's operand - fused with the LayerNorm in front of it where there is one.

`quantize_projections_fp8` runs before LayerNorm folding and claims the linear leaves it can serve:
    layer_norm_wrapper(v, ln) -> {linear_wrapper | linear_cat_wrapper | linear_geglu_wrapper}   ->  linear_fp8_wrapper(v, linears, ln, ...)
    {linear_wrapper | linear_residual_wrapper | linear_geglu_wrapper | linear_cat_wrapper}(v)   ->  linear_fp8_wrapper(v, linears, None, ...)
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import fx, nn

from .. import ops
from .wrappers import (layer_norm_wrapper, linear_cat_wrapper, linear_geglu_wrapper, linear_residual_wrapper,
                       linear_wrapper)


def _fp8_weights(linears):
    """Row-concatenated e4m3 weights, per-channel scales and bf16 bias of the projections; owned by the execution context
    and refreshed in place when a source parameter changes (like the fused bf16 weights, wrappers._cat_weight)."""
    sources = [l.weight for l in linears] + [l.bias for l in linears if l.bias is not None]

    @torch.no_grad()
    def compute():
        w = torch.cat([l.weight.detach() for l in linears], dim=0) if len(linears) > 1 else linears[0].weight.detach()
        wq, ws = ops.quantize_weight_fp8(w)
        if all(l.bias is None for l in linears):
            b = None
        else:
            b = torch.cat([l.bias.detach() if l.bias is not None else
                           torch.zeros(l.out_features, dtype=l.weight.dtype, device=w.device) for l in linears]).to(torch.bfloat16)
        return (wq, ws, b)

    key = ("fp8",) + tuple(id(l) for l in linears)
    return ops.current_context(linears[0].weight.device).derived_weights(key, sources, compute).value


def linear_fp8_wrapper(v: torch.Tensor, linears, layernorm: Optional[nn.LayerNorm] = None, geglu: bool = False,
                       residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[lin_i(LayerNorm?(v))] concatenated (or the GEGLU of the single projection), + residual, on the fp8 matrix pipe."""
    wq, ws, b = _fp8_weights(tuple(linears))
    ln = None if layernorm is None else (layernorm.weight, layernorm.bias, layernorm.eps)
    return ops.linear_fp8(ops.quantize_fp8(v, layernorm=ln), wq, ws, b, geglu=geglu, residual=residual)


torch.fx.wrap("linear_fp8_wrapper")


def _eligible(mods, name_filter, names) -> bool:
    if any(not isinstance(m, nn.Linear) for m in mods):
        return False
    if any(m.in_features % 128 != 0 or m.out_features % 4 != 0 or m.weight.dtype != torch.bfloat16 for m in mods):
        return False
    return all(name_filter in n for n in names)


def quantize_projections_fp8(gm: fx.GraphModule, name_filter: str = "attentions") -> int:
    """Claim the transformer-block projections (module path contains `name_filter`) for the fp8 path."""
    count = 0
    for n in list(gm.graph.nodes):
        if n.op != "call_function":
            continue
        residual = None
        if n.target is linear_wrapper and n.args[2] is False and len(n.args) == 3:
            attrs, geglu = (n.args[1],), False
        elif n.target is linear_residual_wrapper and len(n.args) == 3:
            attrs, geglu, residual = (n.args[1],), False, n.args[2]
        elif n.target is linear_geglu_wrapper:
            attrs, geglu = (n.args[1],), True
        elif n.target is linear_cat_wrapper:
            attrs, geglu = tuple(n.args[1]), False
        else:
            continue
        mods = [gm.get_submodule(a.target) for a in attrs]
        if not _eligible(mods, name_filter, [a.target for a in attrs]):
            continue
        src, ln_attr = n.args[0], None
        if isinstance(src, fx.Node) and src.op == "call_function" and src.target is layer_norm_wrapper and len(src.users) == 1:
            lnmod = gm.get_submodule(src.args[1].target)
            if len(lnmod.normalized_shape) == 1 and lnmod.normalized_shape[0] == mods[0].in_features:
                ln_attr, src_in = src.args[1], src.args[0]
        with gm.graph.inserting_before(n):
            new = gm.graph.call_function(linear_fp8_wrapper, (src_in if ln_attr is not None else src, attrs, ln_attr, geglu, residual))
        n.replace_all_uses_with(new)
        gm.graph.erase_node(n)
        if ln_attr is not None:
            gm.graph.erase_node(src)
        count += 1
    if count:
        gm.graph.eliminate_dead_code()
        gm.graph.lint()
        gm.recompile()
    return count
