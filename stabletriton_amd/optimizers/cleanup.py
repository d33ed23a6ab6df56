"""Small graph clean-ups (additions):

* `dedupe_pure_calls`: identical stateless module calls on the same input are computed once
  (every ResBlock recomputes `SiLU(temb)`, unet_pt.py:81 - 17 identical launches per step);
* `fuse_token_residual`: the spatial transformer's closing `proj_out(...) -> image -> + x`
  (unet_pt.py:236-243) becomes the residual epilogue of the proj_out GEMM (x viewed as tokens).
"""
import operator

import torch
from torch import fx, nn

from .wrappers import linear_residual_wrapper, linear_wrapper

_PURE = (nn.SiLU, nn.GELU, nn.ReLU, nn.Identity)


def dedupe_pure_calls(gm: fx.GraphModule) -> int:
    seen = {}
    count = 0
    for n in list(gm.graph.nodes):
        if n.op == "call_module" and isinstance(gm.get_submodule(n.target), _PURE) and len(n.args) == 1 and not n.kwargs:
            key = (type(gm.get_submodule(n.target)), n.args[0])
            if key in seen:
                n.replace_all_uses_with(seen[key])
                gm.graph.erase_node(n)
                count += 1
            else:
                seen[key] = n
    if count:
        gm.recompile()
    return count


def _is_method(n, name):
    return isinstance(n, fx.Node) and n.op == "call_method" and n.target == name


def fuse_token_residual(gm: fx.GraphModule) -> int:
    """add(contiguous(permute(reshape(linear_wrapper(v, lin, False), b,h,w,c), 0,3,1,2)), x)
    -> same chain on linear_residual_wrapper(v, lin, x.permute(0,2,3,1).reshape(b, h*w, c))."""
    count = 0
    for n in list(gm.graph.nodes):
        if not (n.op == "call_function" and n.target is operator.add and len(n.args) == 2):
            continue
        for img, x in (n.args, n.args[::-1]):
            if not (_is_method(img, "contiguous") and len(img.users) == 1):
                continue
            perm = img.args[0]
            if not (_is_method(perm, "permute") and tuple(perm.args[1:]) == (0, 3, 1, 2) and len(perm.users) == 1):
                continue
            resh = perm.args[0]
            if not (_is_method(resh, "reshape") and len(resh.args) == 5 and len(resh.users) == 1):
                continue
            lin = resh.args[0]
            if not (isinstance(lin, fx.Node) and lin.op == "call_function" and lin.target is linear_wrapper
                    and len(lin.args) == 3 and lin.args[2] is False and len(lin.users) == 1):
                continue
            if not isinstance(x, fx.Node):
                continue
            b, h, w, c = resh.args[1:]
            with gm.graph.inserting_before(lin):
                xp = gm.graph.call_method("permute", (x, 0, 2, 3, 1))
                hw = gm.graph.call_function(operator.mul, (h, w))
                xt = gm.graph.call_method("reshape", (xp, b, hw, c))
                fused = gm.graph.call_function(linear_residual_wrapper, (lin.args[0], lin.args[1], xt))
            lin.replace_all_uses_with(fused)
            gm.graph.erase_node(lin)
            n.replace_all_uses_with(img)
            gm.graph.erase_node(n)
            count += 1
            break
    if count:
        gm.graph.lint()
        gm.recompile()
    return count
