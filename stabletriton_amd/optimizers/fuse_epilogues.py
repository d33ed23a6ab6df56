"""Epilogue fusions on the already-swapped graph (additions; the reference lists
these as intended-but-unbuilt fusions in optimizations.txt:13-66):

* GEGLU into its projection GEMM: chunk(linear(x)) -> a*gelu(g) becomes one
  `linear_geglu_wrapper` (removes 2.5 GB/step of traffic);
* residual adds into the producing GEMM / conv (`attn out-proj + x`,
  `ff out + x`, `x + conv2(h)` of unet_pt.py:93,194,203,209);
* the time-embedding broadcast add into conv1 (`h + temb[:, :, None, None]`,
  unet_pt.py:82-83) as a per-image row bias.
"""
import operator

import torch
from torch import fx

from .wrappers import (conv2d_wrapper, geglu_triton, linear_geglu_wrapper, linear_residual_wrapper,
                       linear_wrapper)


def _is_call(n, fn) -> bool:
    return isinstance(n, fx.Node) and n.op == "call_function" and n.target is fn


def _before(a: fx.Node, b: fx.Node, order) -> bool:
    return order[a] < order[b]


def fuse_geglu_into_linear(gm: fx.GraphModule) -> int:
    count = 0
    for n in list(gm.graph.nodes):
        if not _is_call(n, geglu_triton):
            continue
        st, gt = n.args
        if not (_is_call(st, operator.getitem) and _is_call(gt, operator.getitem)):
            continue
        ch = st.args[0]
        if ch is not gt.args[0] or st.args[1] != 0 or gt.args[1] != 1:
            continue
        if not (isinstance(ch, fx.Node) and ch.op == "call_method" and ch.target == "chunk"):
            continue
        dim = ch.kwargs.get("dim", ch.args[2] if len(ch.args) > 2 else 0)
        if ch.args[1] != 2 or dim not in (-1,):
            continue
        lin = ch.args[0]
        if not (_is_call(lin, linear_wrapper) and lin.args[2] is False):
            continue
        if len(lin.users) != 1 or len(ch.users) != 2 or len(st.users) != 1 or len(gt.users) != 1:
            continue
        with gm.graph.inserting_before(n):
            new = gm.graph.call_function(linear_geglu_wrapper, (lin.args[0], lin.args[1]))
        n.replace_all_uses_with(new)
        for dead in (n, st, gt, ch, lin):
            gm.graph.erase_node(dead)
        count += 1
    gm.recompile()
    return count


def fuse_residual_adds(gm: fx.GraphModule) -> int:
    """add(a, b) where one side is a single-use plain linear_wrapper / conv2d_wrapper."""
    count = 0
    order = {n: i for i, n in enumerate(gm.graph.nodes)}
    for n in list(gm.graph.nodes):
        if not (_is_call(n, operator.add) or _is_call(n, torch.add)) or len(n.args) != 2 or n.kwargs:
            continue
        a, b = n.args
        for prod, other in ((a, b), (b, a)):
            if not isinstance(prod, fx.Node) or not isinstance(other, fx.Node) or len(prod.users) != 1:
                continue
            if _is_call(prod, linear_wrapper) and prod.args[2] is False:
                with gm.graph.inserting_before(n):
                    new = gm.graph.call_function(linear_residual_wrapper, (prod.args[0], prod.args[1], other))
            elif _is_call(prod, conv2d_wrapper) and (len(prod.args) < 5 or prod.args[4] is None) \
                    and "residual" not in prod.kwargs:
                args = list(prod.args) + [False, None, None][len(prod.args) - 2:]
                args[4] = other
                with gm.graph.inserting_before(n):
                    new = gm.graph.call_function(conv2d_wrapper, tuple(args))
            else:
                continue
            n.replace_all_uses_with(new)
            gm.graph.erase_node(n)
            gm.graph.erase_node(prod)
            count += 1
            break
    gm.recompile()
    return count


def fuse_temb_add(gm: fx.GraphModule) -> int:
    """conv(h) + temb[:, :, None, None]  ->  conv with per-image row bias."""
    count = 0
    for n in list(gm.graph.nodes):
        if not _is_call(n, operator.add) or len(n.args) != 2:
            continue
        a, b = n.args
        for conv, gi in ((a, b), (b, a)):
            if not (_is_call(conv, conv2d_wrapper) and _is_call(gi, operator.getitem)):
                continue
            idx = gi.args[1]
            if not (isinstance(idx, tuple) and len(idx) == 4 and idx[0] == slice(None) and idx[1] == slice(None)
                    and idx[2] is None and idx[3] is None):
                continue
            if len(conv.users) != 1 or (len(conv.args) > 3 and conv.args[3] is not None):
                continue
            args = list(conv.args) + [False, None, None][len(conv.args) - 2:]
            args[3] = gi.args[0]
            with gm.graph.inserting_before(n):
                new = gm.graph.call_function(conv2d_wrapper, tuple(args))
            n.replace_all_uses_with(new)
            gm.graph.erase_node(n)
            gm.graph.erase_node(conv)
            if len(gi.users) == 0:
                gm.graph.erase_node(gi)
            count += 1
            break
    gm.recompile()
    return count
