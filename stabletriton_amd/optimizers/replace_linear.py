"""Linear and activation(Linear) call_module pairs -> `linear_wrapper`.
Counterpart of reference optimizers/replace_linear.py:40-77.  The reference
keeps the plain-Linear pass disabled because its Triton GEMM lost to cuBLAS
(optimization.py:18-20); here the MFMA GEMM is the product, so it is on."""
from typing import Callable

from torch import fx, nn

from ..fx_match import replace_pattern
from .wrappers import linear_wrapper


def replace_linear(gm: fx.GraphModule) -> int:
    class Pattern(nn.Module):
        def __init__(self):
            super().__init__()
            self.linear = nn.Linear(1, 1)

        def forward(self, v):
            return self.linear(v)

    return len(replace_pattern(
        gm, Pattern(), lambda g, m: g.call_function(linear_wrapper, (m.bindings["v"], g.get_attr(m.modules["linear"]), False))))


def replace_linear_activ(gm: fx.GraphModule, activation: Callable) -> int:
    if not isinstance(activation, nn.SiLU):
        raise NotImplementedError("the GEMM epilogue implements SiLU only")

    class Pattern(nn.Module):
        def __init__(self):
            super().__init__()
            self.linear = nn.Linear(1, 1)
            self.activation = activation

        def forward(self, v):
            return self.activation(self.linear(v))

    return len(replace_pattern(
        gm, Pattern(), lambda g, m: g.call_function(linear_wrapper, (m.bindings["v"], g.get_attr(m.modules["linear"]), True))))
