"""Fuse the eager attention core into one `attention_wrapper` call.

Pattern = the canonical eager form of unet_pt.py:133-142 (view/transpose ->
matmul * scale -> softmax -> matmul -> transpose/contiguous/view), the same
sub-graph the reference rewrites in optimizers/replace_attention.py:74-94.
The reference hands the un-split (B,T,C) tensors to xformers, which is only
correct for one head (defect 1); here the wrapper receives
num_heads/head_dim and the kernel indexes heads inside the projection layout.
"""
import torch
from torch import fx

from ..fx_match import replace_pattern
from .wrappers import attention_wrapper


def _pattern(q, k, v, sm_scale, num_heads, head_dim):
    b, t, c = q.size()
    q = q.view(q.size(0), q.size(1), num_heads, head_dim).transpose(1, 2)
    k = k.view(k.size(0), k.size(1), num_heads, head_dim).transpose(1, 2)
    v = v.view(v.size(0), v.size(1), num_heads, head_dim).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) * sm_scale
    attn = torch.softmax(scores, dim=-1)
    out = torch.matmul(attn, v)
    return out.transpose(1, 2).contiguous().view(b, t, c)


def fuse_attention(gm: fx.GraphModule) -> int:
    def build(graph: fx.Graph, m):
        b = m.bindings
        return graph.call_function(attention_wrapper,
                                   (b["q"], b["k"], b["v"], None, b["sm_scale"], b["num_heads"], b["head_dim"]))

    return len(replace_pattern(gm, _pattern, build))
