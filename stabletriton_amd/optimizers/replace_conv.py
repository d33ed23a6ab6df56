"""Conv2d call_module -> `conv2d_wrapper` (NHWC implicit-GEMM kernel).

No reference counterpart: the reference leaves all 51 convolutions to cuDNN
(optimizations.txt:5; its Triton conv experiments are dead code,
kernels/Conv_Kernels/README.md:2).  BASELINE.json's north_star puts the
resnet-block convolutions on the hot path.
`nearest-2x interpolate -> conv` (unet_pt.py:264-266) folds into the gather.
"""
import torch
from torch import fx, nn

from .wrappers import conv2d_wrapper


def _supported(conv: nn.Conv2d) -> bool:
    return (conv.groups == 1 and conv.dilation == (1, 1) and not isinstance(conv.padding, str)
            and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
            and conv.padding_mode == "zeros")


def _is_nearest_2x(n: fx.Node) -> bool:
    if n.op != "call_function" or n.target is not torch.nn.functional.interpolate:
        return False
    kw = dict(n.kwargs)
    sf = kw.get("scale_factor", n.args[2] if len(n.args) > 2 else None)
    mode = kw.get("mode", n.args[3] if len(n.args) > 3 else "nearest")
    size = kw.get("size", n.args[1] if len(n.args) > 1 else None)
    return size is None and sf in (2, 2.0) and mode == "nearest"


def replace_conv(gm: fx.GraphModule) -> int:
    mods = dict(gm.named_modules())
    count = 0
    for n in list(gm.graph.nodes):
        if n.op != "call_module" or not isinstance(mods.get(n.target), nn.Conv2d) or not _supported(mods[n.target]):
            continue
        src = n.args[0]
        ups = False
        if isinstance(src, fx.Node) and _is_nearest_2x(src) and len(src.users) == 1:
            ups, src_in = True, src.args[0]
        with gm.graph.inserting_before(n):
            new = gm.graph.call_function(conv2d_wrapper, (src_in if ups else src, gm.graph.get_attr(n.target), ups))
        n.replace_all_uses_with(new)
        gm.graph.erase_node(n)
        if ups:
            gm.graph.erase_node(src)
        count += 1
    gm.graph.lint()
    gm.recompile()
    return count
