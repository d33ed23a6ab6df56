"""GroupNorm and activation(GroupNorm) call_module pairs -> `group_norm_wrapper`.
Counterpart of reference optimizers/replace_groupnorm.py:23-61 (module matched
by type; the wrapper is handed the original module)."""
from typing import Callable

import torch
from torch import fx, nn

from ..fx_match import replace_pattern
from .wrappers import group_norm_wrapper


def replace_group_norm(gm: fx.GraphModule) -> int:
    class Pattern(nn.Module):
        def __init__(self):
            super().__init__()
            self.groupnorm = nn.GroupNorm(1, 1)

        def forward(self, v):
            return self.groupnorm(v)

    return len(replace_pattern(
        gm, Pattern(),
        lambda g, m: g.call_function(group_norm_wrapper, (m.bindings["v"], g.get_attr(m.modules["groupnorm"]), False))))


def replace_group_norm_activation(gm: fx.GraphModule, activation: Callable) -> int:
    if not isinstance(activation, nn.SiLU):
        raise NotImplementedError("the fused GroupNorm kernel implements SiLU only")

    class Pattern(nn.Module):
        def __init__(self):
            super().__init__()
            self.groupnorm = nn.GroupNorm(1, 1)
            self.activation = activation

        def forward(self, v):
            return self.activation(self.groupnorm(v))

    return len(replace_pattern(
        gm, Pattern(),
        lambda g, m: g.call_function(group_norm_wrapper, (m.bindings["v"], g.get_attr(m.modules["groupnorm"]), True))))
