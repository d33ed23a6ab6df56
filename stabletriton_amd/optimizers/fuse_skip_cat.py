"""Skip connections without the concatenated tensor (addition).

The decoder concatenates the running activation with a skip tensor along the channels (unet_pt.py:352-357) and hands the
result to a resnet, where it has exactly two readers: norm1 (GroupNorm, already fed by the statistics of both producers)
and the 1x1 shortcut conv.  Both kernels can read the two halves where they lie: `group_norm_stats_wrapper` takes the
pair, `conv2d_cat_wrapper` replaces the shortcut, and the torch.cat node - nine copies of up to 21 MB per step - goes away.
Runs after fuse_groupnorm_stats; a concatenation with any other reader is left alone.
"""
from __future__ import annotations

import torch
from torch import fx

from .wrappers import conv2d_cat_wrapper, conv2d_wrapper, group_norm_stats_wrapper


def _is_fn(n, fn) -> bool:
    return isinstance(n, fx.Node) and n.op == "call_function" and n.target is fn


def _plain_1x1(gm: fx.GraphModule, n: fx.Node) -> bool:
    """conv2d_wrapper(cat, conv[, upsample2x=False, rowbias=None, residual=..., emit_colstats=...]) with a 1x1 / stride 1 / pad 0 conv."""
    if len(n.args) > 2 and n.args[2]:
        return False
    if n.kwargs.get("upsample2x") or n.kwargs.get("rowbias") is not None or (len(n.args) > 3 and n.args[3] is not None):
        return False
    conv = n.args[1]
    if isinstance(conv, fx.Node):
        if conv.op != "get_attr":
            return False
        try:
            conv = gm.get_submodule(conv.target)
        except AttributeError:
            return False
    if not isinstance(conv, torch.nn.Conv2d):
        return False
    return conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1


def fuse_skip_cat(gm: fx.GraphModule) -> int:
    count = 0
    for n in list(gm.graph.nodes):
        if not _is_fn(n, torch.cat):
            continue
        parts = n.args[0]
        dim = n.kwargs.get("dim", n.args[1] if len(n.args) > 1 else 0)
        if dim != 1 or not isinstance(parts, (list, tuple)) or len(parts) != 2 or not n.users:
            continue
        users = list(n.users)
        ok = True
        for u in users:
            if _is_fn(u, group_norm_stats_wrapper) and u.args[0] is n and len(u.args[1]) == 2:
                continue
            if _is_fn(u, conv2d_wrapper) and u.args[0] is n and _plain_1x1(gm, u):
                continue
            ok = False
        if not ok:
            continue
        a, b = parts
        for u in users:
            if _is_fn(u, group_norm_stats_wrapper):
                u.args = ((a, b),) + tuple(u.args[1:])
            else:
                residual = u.kwargs.get("residual", u.args[4] if len(u.args) > 4 else None)
                emit = u.kwargs.get("emit_colstats", u.args[5] if len(u.args) > 5 else False)
                with gm.graph.inserting_before(u):
                    new = gm.graph.call_function(conv2d_cat_wrapper, (a, b, u.args[1]), {"residual": residual, "emit_colstats": emit})
                u.replace_all_uses_with(new)
                gm.graph.erase_node(u)
        gm.graph.erase_node(n)
        count += 1
    if count:
        gm.graph.lint()
        gm.recompile()
    return count
