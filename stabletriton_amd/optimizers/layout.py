"""Keep image tensors channels-last (NHWC) across the transformer boundary.

The eager model goes NCHW -> tokens -> NCHW around every spatial transformer
(unet_pt.py:228-241).  With every image op producing channels_last tensors the
`permute(0,2,3,1).reshape(b,hw,c)` is already a view; the way back,
`reshape(b,h,w,c).permute(0,3,1,2).contiguous()`, would copy into NCHW.  This
pass asks that `contiguous` for channels_last instead, which makes it a no-op
and keeps the residual add and everything after it in NHWC.
"""
import torch
from torch import fx


def keep_channels_last(gm: fx.GraphModule) -> int:
    count = 0
    for n in gm.graph.nodes:
        if n.op == "call_method" and n.target == "contiguous" and len(n.args) == 1 and not n.kwargs:
            src = n.args[0]
            if (isinstance(src, fx.Node) and src.op == "call_method" and src.target == "permute"
                    and tuple(src.args[1:]) in ((0, 3, 1, 2), ((0, 3, 1, 2),))):
                n.kwargs = {"memory_format": torch.channels_last}
                count += 1
    gm.recompile()
    return count
