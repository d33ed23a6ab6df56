"""fp8 plan of the compiled graph (addition; SURVEY.md 8f-4, BASELINE config #5).

Seed in the reference: its unused fused-projection kernel stores the projection weights as fp8 and up-converts them before
the product (kernels/attention_proj.py:36-39, 105-155).  Here the three big projections of every transformer block - q|k|v,
the GEGLU projection, the feed-forward output projection: three quarters of the Linear FLOPs - run with BOTH operands in OCP
e4m3 on the block-scaled matrix instruction (twice the bf16 rate), and nothing is launched to quantise:

    producer (bf16 GEMM or fp8 FF2)  --epilogue-->  x (bf16, residual stream) + row statistics + e4m3 copy of x
    q|k|v / GEGLU projection          =  fp8 GEMM on that copy, LayerNorm folded exactly as in the bf16 graph
    GEGLU projection                  --epilogue-->  ONLY the e4m3 copy of its output
    feed-forward output projection    =  fp8 GEMM on that copy (+ residual) --epilogue--> x', statistics, e4m3 copy of x'

The copies carry ONE scale per tensor, taken from the previous denoise step's max |value| (delayed scaling; ops.Fp8Scales):
a floating-point format keeps its relative precision at any scale, so a per-row scale buys range, not accuracy, and range is
what the previous step predicts.  One launch per step (`st_fp8_update_scales`) turns the maxima into the next scales.
The other projections (to_out, the cross-attention query projection with its attention, proj_in / proj_out) are one-round,
latency-bound launches at batch 1 and stay bf16.  The pass runs after the bf16 fusions and rewrites their result.
"""
from __future__ import annotations

import operator
from typing import Optional

import torch
from torch import fx, nn

from .. import ops
from .wrappers import linear_residual_wrapper, linear_wrapper, ln_linear_wrapper


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


def _ln_fold_fp8(layernorm: nn.LayerNorm, linears):
    sources = [layernorm.weight, layernorm.bias] + [l.weight for l in linears] + [l.bias for l in linears if l.bias is not None]

    @torch.no_grad()
    def compute():
        w = torch.cat([l.weight.detach() for l in linears], dim=0) if len(linears) > 1 else linears[0].weight.detach()
        if all(l.bias is None for l in linears):
            b = None
        else:
            b = torch.cat([l.bias.detach().float() if l.bias is not None else
                           torch.zeros(l.out_features, dtype=torch.float32, device=w.device) for l in linears])
        return ops.fold_layer_norm_fp8(layernorm.weight.detach(), layernorm.bias.detach(), w, b)

    key = ("ln8", id(layernorm)) + tuple(id(l) for l in linears)
    return ops.current_context(layernorm.weight.device).derived_weights(key, sources, compute).value


def linear_emit8_wrapper(v: torch.Tensor, linear: nn.Linear, residual: Optional[torch.Tensor], emit_stats: bool, site: int):
    """linear_wrapper / linear_residual_wrapper whose epilogue also leaves the e4m3 copy of the output: (out, stats, Fp8Act)."""
    r = ops.linear(v, linear.weight, linear.bias, residual=residual, emit_stats=emit_stats, emit_q8=("site", site))
    return r if emit_stats else (r[0], None, r[1])


def ln_linear_fp8_wrapper(x8, stats, layernorm: nn.LayerNorm, linears, geglu: bool = False, site: Optional[int] = None):
    """ln_linear_wrapper on the fp8 matrix pipe: `x8` is the e4m3 copy of the un-normalised input its producer left.
    With `site` (the GEGLU projection in front of an fp8 feed-forward output projection) ONLY the e4m3 copy of the result is
    produced and returned."""
    wq, ws, c, d = _ln_fold_fp8(layernorm, tuple(linears))
    return ops.linear_fp8x(x8, wq, ws, None, geglu=geglu, ln=(stats, c, d, layernorm.eps),
                           emit_q8=None if site is None else ("site", site), want_out=site is None)


def linear_fp8_residual_wrapper(x8, linear: nn.Linear, residual: torch.Tensor, emit_stats: bool = False, site: Optional[int] = None):
    """linear_residual_wrapper on the fp8 matrix pipe (the feed-forward output projection): x8 is the e4m3 copy the GEGLU
    projection left.  Result like the wrapper it replaces: out, or (out, stats); with `site` (out, stats or None, Fp8Act)."""
    wq, ws, b = _fp8_weights((linear,))
    r = ops.linear_fp8x(x8, wq, ws, b, residual=residual, emit_stats=emit_stats, emit_q8=None if site is None else ("site", site))
    if site is not None and not emit_stats:
        return r[0], None, r[1]
    return r


for _name in ("linear_emit8_wrapper", "ln_linear_fp8_wrapper", "linear_fp8_residual_wrapper"):
    torch.fx.wrap(_name)


def _is(n, fn) -> bool:
    return isinstance(n, fx.Node) and n.op == "call_function" and n.target is fn


def _lin_ok(m, name, name_filter) -> bool:
    return isinstance(m, nn.Linear) and m.in_features % 128 == 0 and m.out_features % 8 == 0 and m.weight.dtype == torch.bfloat16 and name_filter in name


def plan_fp8(gm: fx.GraphModule, name_filter: str = "attentions") -> dict:
    """Rewrite the bf16 graph (after LayerNorm folding) to the fp8 plan.  Returns the counts by role."""
    g = gm.graph
    sites = [0]

    def new_site() -> int:
        sites[0] += 1
        return sites[0]

    stats = {"ln_projections": 0, "ff_out_projections": 0, "emitting_producers": 0}
    # ---- A: LayerNorm-folded projections take the e4m3 copy their producer leaves ----------------------------------------
    for L in list(g.nodes):
        if not _is(L, ln_linear_wrapper):
            continue
        x0, st, ln_attr, linears = L.args[0], L.args[1], L.args[2], tuple(L.args[3])
        geglu = L.args[4] if len(L.args) > 4 else L.kwargs.get("geglu", False)
        mods = [gm.get_submodule(a.target) for a in linears]
        if not all(_lin_ok(m, a.target, name_filter) for m, a in zip(mods, linears)):
            continue
        if not (_is(x0, operator.getitem) and x0.args[1] == 0 and _is(st, operator.getitem) and st.args[0] is x0.args[0]):
            continue
        P = x0.args[0]
        if _is(P, linear_emit8_wrapper):
            q8n = next(u for u in P.users if _is(u, operator.getitem) and u.args[1] == 2)
        else:
            if _is(P, linear_wrapper) and len(P.args) == 4 and P.args[2] is False and P.args[3] is True and not P.kwargs:
                v, lin, res = P.args[0], P.args[1], None
            elif _is(P, linear_residual_wrapper) and len(P.args) == 4 and P.args[3] is True and not P.kwargs:
                v, lin, res = P.args[0], P.args[1], P.args[2]
            else:
                continue
            if gm.get_submodule(lin.target).out_features % 8 != 0:
                continue
            with g.inserting_before(P):
                Pn = g.call_function(linear_emit8_wrapper, (v, lin, res, True, new_site()))
            P.replace_all_uses_with(Pn)
            g.erase_node(P)
            with g.inserting_after(Pn):
                q8n = g.call_function(operator.getitem, (Pn, 2))
            stats["emitting_producers"] += 1
        with g.inserting_before(L):
            new = g.call_function(ln_linear_fp8_wrapper, (q8n, st, ln_attr, linears, bool(geglu), None))
        L.replace_all_uses_with(new)
        g.erase_node(L)
        stats["ln_projections"] += 1
    # ---- B: GEGLU projection -> feed-forward output projection stays in e4m3 -------------------------------------------
    for F1 in list(g.nodes):
        if not (_is(F1, ln_linear_fp8_wrapper) and F1.args[4] is True and len(F1.users) == 1):
            continue
        F2 = next(iter(F1.users))
        if _is(F2, linear_residual_wrapper) and F2.args[0] is F1 and not F2.kwargs and len(F2.args) in (3, 4):
            lin, res, emit, site = F2.args[1], F2.args[2], (len(F2.args) == 4 and F2.args[3] is True), None
        elif _is(F2, linear_emit8_wrapper) and F2.args[0] is F1:
            lin, res, emit, site = F2.args[1], F2.args[2], F2.args[3], F2.args[4]
            if res is None:
                continue
        else:
            continue
        if not _lin_ok(gm.get_submodule(lin.target), lin.target, name_filter):
            continue
        F1.args = tuple(F1.args[:5]) + (new_site(),)
        with g.inserting_before(F2):
            new = g.call_function(linear_fp8_residual_wrapper, (F1, lin, res, bool(emit), site))
        F2.replace_all_uses_with(new)
        g.erase_node(F2)
        stats["ff_out_projections"] += 1
    g.eliminate_dead_code()
    g.lint()
    gm.recompile()
    stats["e4m3_tensors"] = sites[0]
    return stats
