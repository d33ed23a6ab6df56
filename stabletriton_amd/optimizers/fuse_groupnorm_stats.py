"""GroupNorm statistics from the producing launch (addition).

Every GroupNorm of the UNet reads what a conv or a GEMM epilogue has just written (conv1 -> norm2, conv2 + residual ->
the next block's norm1, proj_out + residual -> the next resnet, unet_pt.py:74-95,236-243) or the channel concatenation
of two such tensors (the decoder's skip connections, unet_pt.py:352-357).  The producer's epilogue holds every output
value in registers, so it also emits per-channel partial sums (`emit_colstats`); `group_norm_stats_wrapper` then needs
only a tiny finalize launch and the apply pass - the statistics launch and its read of the whole tensor disappear.
Producers that cannot emit (conv_in's thin kernel, ragged shapes) return no statistics and the wrapper falls back to the
three-launch GroupNorm at run time.
"""
from __future__ import annotations

import operator
from typing import List, Optional

import torch
from torch import fx

from .wrappers import conv2d_wrapper, group_norm_stats_wrapper, group_norm_wrapper, linear_residual_wrapper

_VIEWS = ("reshape", "view", "permute", "contiguous")


def _is_fn(n, fn) -> bool:
    return isinstance(n, fx.Node) and n.op == "call_function" and n.target is fn


def _root(v: fx.Node) -> fx.Node:
    while isinstance(v, fx.Node) and v.op == "call_method" and v.target in _VIEWS:
        v = v.args[0]
    return v


def _can_emit(n: fx.Node) -> bool:
    if _is_fn(n, conv2d_wrapper):
        return True
    if _is_fn(n, linear_residual_wrapper):
        return not (len(n.args) > 3 and n.args[3]) and not n.kwargs.get("emit_stats")
    return False


def _producers(v: fx.Node) -> Optional[List[fx.Node]]:
    """Producer nodes of the channel ranges of v, in channel order; None when some range has no emitting producer."""
    v = _root(v)
    if _is_fn(v, operator.getitem) and v.args[1] == 0 and _can_emit(v.args[0]) and v.args[0].kwargs.get("emit_colstats"):
        return [v.args[0]]
    if _can_emit(v):
        return [v]
    if _is_fn(v, torch.cat):
        parts = v.args[0]
        dim = v.kwargs.get("dim", v.args[1] if len(v.args) > 1 else 0)
        if dim == 1 and isinstance(parts, (list, tuple)) and len(parts) == 2:
            a, b = _producers(parts[0]), _producers(parts[1])
            if a is not None and b is not None and len(a) == 1 and len(b) == 1:
                return a + b
    return None


def _stats_of(gm: fx.GraphModule, prod: fx.Node) -> fx.Node:
    """The statistics output of a producer, switching it to (out, stats) form on first use."""
    if not prod.kwargs.get("emit_colstats"):
        prod.kwargs = {**prod.kwargs, "emit_colstats": True}
        with gm.graph.inserting_after(prod):
            st = gm.graph.call_function(operator.getitem, (prod, 1))
            out = gm.graph.call_function(operator.getitem, (prod, 0))
        prod.replace_all_uses_with(out, delete_user_cb=lambda u: u is not out and u is not st)
        return st
    for u in prod.users:
        if _is_fn(u, operator.getitem) and u.args[1] == 1:
            return u
    raise RuntimeError("producer in statistics form without a statistics output")


def fuse_groupnorm_stats(gm: fx.GraphModule) -> int:
    count = 0
    for n in list(gm.graph.nodes):
        if not _is_fn(n, group_norm_wrapper):
            continue
        prods = _producers(n.args[0])
        if prods is None:
            continue
        stats = tuple(_stats_of(gm, p) for p in prods)
        with gm.graph.inserting_before(n):
            new = gm.graph.call_function(group_norm_stats_wrapper, (n.args[0], stats, n.args[1], n.args[2]))
        n.replace_all_uses_with(new)
        gm.graph.erase_node(n)
        count += 1
    if count:
        gm.graph.lint()
        gm.recompile()
    return count
