"""Erase inference-time Dropout (p = 0) nodes.
Counterpart of reference optimizers/remove_dropout.py:19-33."""
import torch
from torch import fx, nn


def remove_dropout(gm: fx.GraphModule) -> int:
    mods = dict(gm.named_modules())
    n_removed = 0
    for n in list(gm.graph.nodes):
        is_mod = n.op == "call_module" and isinstance(mods.get(n.target), nn.Dropout)
        is_fn = n.op == "call_function" and n.target is torch.nn.functional.dropout
        if not (is_mod or is_fn):
            continue
        if is_mod and mods[n.target].training and mods[n.target].p > 0:
            continue                       # a live training-mode dropout is not an identity
        n.replace_all_uses_with(n.args[0])
        gm.graph.erase_node(n)
        n_removed += 1
    gm.recompile()
    return n_removed
