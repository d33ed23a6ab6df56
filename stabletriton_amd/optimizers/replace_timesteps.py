"""Sinusoidal timestep embedding -> `timestep_embedding_wrapper` (one small kernel).

Counterpart of reference optimizers/replace_timesteps.py:43-58, whose pattern
never matches the real UNet (0 sites).  This pass matches by
structure instead: cat([cos(e), sin(e)], -1) with
e = t[:, None].float() * exp(<arange(half) expression>)[None, :].
"""
import operator

import torch
from torch import fx

from .wrappers import timestep_embedding_wrapper


def _find_arange_half(n: fx.Node, depth: int = 0):
    if depth > 8 or not isinstance(n, fx.Node):
        return None
    if n.op == "call_function" and n.target is torch.arange:
        a = n.args[0] if n.args else n.kwargs.get("end")
        return a if isinstance(a, int) else None
    for a in n.all_input_nodes:
        r = _find_arange_half(a, depth + 1)
        if r is not None:
            return r
    return None


def fuse_timesteps(gm: fx.GraphModule) -> int:
    count = 0
    for n in list(gm.graph.nodes):
        if not (n.op == "call_function" and n.target is torch.cat):
            continue
        parts = n.args[0]
        dim = n.kwargs.get("dim", n.args[1] if len(n.args) > 1 else 0)
        if not (isinstance(parts, (list, tuple)) and len(parts) == 2 and dim == -1):
            continue
        c, s = parts
        if not (isinstance(c, fx.Node) and isinstance(s, fx.Node) and c.op == s.op == "call_function"
                and c.target is torch.cos and s.target is torch.sin and c.args[0] is s.args[0]):
            continue
        emb = c.args[0]
        if not (emb.op == "call_function" and emb.target is operator.mul):
            continue
        tf, fr = emb.args
        # t[:, None].float()
        if not (isinstance(tf, fx.Node) and tf.op == "call_method" and tf.target == "float"):
            continue
        tg = tf.args[0]
        if not (tg.op == "call_function" and tg.target is operator.getitem and tg.args[1] == (slice(None), None)):
            continue
        half = _find_arange_half(fr)
        if half is None:
            continue
        t = tg.args[0]
        with gm.graph.inserting_before(n):
            new = gm.graph.call_function(timestep_embedding_wrapper, (t, 2 * half))
        n.replace_all_uses_with(new)
        count += 1
    if count:
        gm.graph.eliminate_dead_code()
        gm.recompile()
    return count
