"""LayerNorm call_module -> `layer_norm_wrapper`.
Counterpart of reference optimizers/replace_layernorm.py:30-47."""
from torch import fx, nn

from ..fx_match import replace_pattern
from .wrappers import layer_norm_wrapper


def replace_layer_norm(gm: fx.GraphModule) -> int:
    class Pattern(nn.Module):
        def __init__(self):
            super().__init__()
            self.layernorm = nn.LayerNorm(1)

        def forward(self, v):
            return self.layernorm(v)

    # only last-dimension norms are implemented by the kernel
    ok = lambda name, mod: len(mod.normalized_shape) == 1 and mod.elementwise_affine
    return len(replace_pattern(
        gm, Pattern(),
        lambda g, m: g.call_function(layer_norm_wrapper, (m.bindings["v"], g.get_attr(m.modules["layernorm"]))),
        module_filter=ok))
