"""`torch.library` registration of the HIP operators (last row).

The compiled graph calls the launchers in `ops.py` directly (fx leaf functions -> ctypes -> C ABI).  This module gives
the same operators a dispatcher identity - `torch.ops.st.attention`, `st.group_norm_silu`, `st.geglu`, `st.linear_act`,
`st.conv2d_epilogue`, `st.layer_norm` - each with a Meta (fake) kernel that computes output shape, dtype and layout
without launching, so code that contains them can be traced with FakeTensor / `torch.compile`'s front end or
`make_fx`.  The CUDA (HIP) kernel of every op is the C-ABI launcher; there is no CPU kernel: on CPU tensors the
dispatcher raises (the library has no CPU fallback anywhere).

    import stabletriton_amd.torch_ops          # registers the namespace `st`
    y = torch.ops.st.linear_act(x, w, b, False)
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

_lib = torch.library.Library("st", "DEF")
_lib.define("attention(Tensor q, Tensor k, Tensor v, int num_heads, float scale, int head_dim=64) -> Tensor")      # head_dim: 16 / 32 / 64 / 128 (st_attention)
_lib.define("group_norm_silu(Tensor x, int num_groups, Tensor weight, Tensor bias, float eps, bool silu) -> Tensor")
_lib.define("geglu(Tensor state, Tensor gate) -> Tensor")
_lib.define("linear_act(Tensor x, Tensor weight, Tensor? bias, bool silu, bool geglu=False, Tensor? residual=None) -> Tensor")
_lib.define("conv2d_epilogue(Tensor x, Tensor weight, Tensor? bias, int stride, int padding, bool upsample2x=False, "
            "Tensor? rowbias=None, Tensor? residual=None) -> Tensor")
_lib.define("layer_norm(Tensor x, Tensor weight, Tensor bias, float eps) -> Tensor")


# ---- HIP kernels: the C-ABI launchers --------------------------------------------------------------
ATTENTION_HEAD_DIMS = (16, 32, 64, 128)      # what st_attention takes (include/stabletriton_amd.h; kernels/attention_fa2.py:118-123)


def _check_attention(q, k, v, num_heads, head_dim):
    torch._check(q.dim() == 3 and k.dim() == 3 and v.dim() == 3, lambda: "attention expects (B, T, H*D) tensors")
    torch._check(head_dim in ATTENTION_HEAD_DIMS, lambda: f"attention: head_dim {head_dim} not in {ATTENTION_HEAD_DIMS}")
    torch._check(q.shape[-1] == num_heads * head_dim and k.shape[-1] == q.shape[-1] and v.shape == k.shape,
                 lambda: f"attention: the last dimension must be num_heads * head_dim = {num_heads * head_dim} on q, k and v")


def _attention(q, k, v, num_heads, scale, head_dim=64):
    _check_attention(q, k, v, num_heads, head_dim)
    return ops.attention(q, k, v, num_heads, scale)


def _group_norm_silu(x, num_groups, weight, bias, eps, silu):
    return ops.group_norm(x, num_groups, weight, bias, eps, silu)


def _geglu(state, gate):
    return ops.geglu(state, gate)


def _linear_act(x, weight, bias, silu, geglu=False, residual=None):
    return ops.linear(x, weight, bias, silu=silu, geglu=geglu, residual=residual)


def _conv2d_epilogue(x, weight, bias, stride, padding, upsample2x=False, rowbias=None, residual=None):
    return ops.conv2d(x, weight, bias, stride, padding, upsample2x=upsample2x, rowbias=rowbias, residual=residual)


def _layer_norm(x, weight, bias, eps):
    return ops.layer_norm(x, weight, bias, eps)


for _name, _fn in (("attention", _attention), ("group_norm_silu", _group_norm_silu), ("geglu", _geglu),
                   ("linear_act", _linear_act), ("conv2d_epilogue", _conv2d_epilogue), ("layer_norm", _layer_norm)):
    _lib.impl(_name, _fn, "CUDA")


# ---- Meta kernels: shapes, dtypes, layouts ---------------------------------------------------------
def _meta_attention(q, k, v, num_heads, scale, head_dim=64):
    _check_attention(q, k, v, num_heads, head_dim)
    return q.new_empty(q.shape)


def _meta_like(x, *rest):
    return torch.empty_like(x)            # preserves channels_last


def _meta_geglu(state, gate):
    torch._check(state.shape == gate.shape, lambda: "geglu: state and gate must have the same shape")
    return state.new_empty(state.shape)


def _meta_linear_act(x, weight, bias, silu, geglu=False, residual=None):
    torch._check(weight.dim() == 2 and weight.shape[1] == x.shape[-1], lambda: "linear: weight does not match input K")
    n = weight.shape[0] // 2 if geglu else weight.shape[0]
    return x.new_empty((*x.shape[:-1], n))


def _meta_conv2d(x, weight, bias, stride, padding, upsample2x=False, rowbias=None, residual=None):
    torch._check(x.dim() == 4 and weight.dim() == 4, lambda: "conv2d expects 4-D input and weight")
    n, _, h, w = x.shape
    cout, _, r, s = weight.shape
    he, we = (2 * h, 2 * w) if upsample2x else (h, w)
    ho, wo = (he + 2 * padding - r) // stride + 1, (we + 2 * padding - s) // stride + 1
    return torch.empty((n, cout, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)


_lib.impl("attention", _meta_attention, "Meta")
_lib.impl("group_norm_silu", _meta_like, "Meta")
_lib.impl("geglu", _meta_geglu, "Meta")
_lib.impl("linear_act", _meta_linear_act, "Meta")
_lib.impl("conv2d_epilogue", _meta_conv2d, "Meta")
_lib.impl("layer_norm", _meta_like, "Meta")
