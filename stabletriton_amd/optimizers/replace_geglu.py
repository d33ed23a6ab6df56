"""`state * gelu(gate)` -> `geglu_triton(state, gate)`.
Counterpart of reference optimizers/replace_geglu.py:33-41; the reference
forces `.contiguous()` copies of both halves (:38), this kernel takes the
strided chunk views as they are."""
import torch
from torch import fx

from ..fx_match import replace_pattern
from .wrappers import geglu_triton


def _pattern(state, gate):
    return state * torch.nn.functional.gelu(gate)


def fuse_geglu(gm: fx.GraphModule) -> int:
    return len(replace_pattern(
        gm, _pattern, lambda graph, m: graph.call_function(geglu_triton, (m.bindings["state"], m.bindings["gate"]))))
