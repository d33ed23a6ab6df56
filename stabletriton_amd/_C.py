"""ctypes binding of include/stabletriton_amd.h.

There is no fallback: if the hipcc-built library is missing or an entry point
rejects its arguments, the op raises.  (The product path never computes on the
CPU and never imports oracle/.)
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads torch's libamdhip64 first so ours binds to the same runtime)

from .build import lib_path

ST_F32, ST_BF16, ST_F16, ST_F32S = 0, 1, 2, 3
ST_NCHW, ST_NHWC = 0, 1
EPI_BIAS, EPI_SILU, EPI_GEGLU, EPI_RESIDUAL, EPI_ROWBIAS = 1, 2, 4, 8, 16
ABI_VERSION = 16

_p, _i, _l, _f, _z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol declared in the header
SIGNATURES = {
    "st_abi_version": (_i, []),
    "st_last_error": (C.c_char_p, []),
    "st_group_norm_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "st_group_norm": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _i, _p, _p]),
    "st_layer_norm": (_i, [_p, _p, _p, _p, _i, _i, _f, _i, _p]),
    "st_geglu": (_i, [_p, _p, _p, _i, _i, _l, _l, _l, _i, _p]),
    "st_linear": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _p, _z, _p, _i, _p, _p, _i, _p, _p, _z, _p]),
    "st_ln_linear": (_i, [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _l, _l, _f, _i, _i, _p, _z, _p]),
    "st_ln_linear_xattn": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _l, _l, _f, _i, _i, _i, _l, _l, _f, _i, _p, _z, _p]),
    "st_attention": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _l, _l, _f, _i, _p]),
    "st_conv2d": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _i, _p, _p, _z, _p]),
    "st_group_norm_from_stats": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p, _i, _i, _p, _i, _i, _p, _p]),
    "st_group_norm_from_stats_cat": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p, _i, _i, _p, _i, _i, _p, _p]),
    "st_conv1x1_cat": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p, _i, _p, _p, _z, _p]),
    "st_euler_step": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _i, _p]),
    "st_step_advance": (_i, [_p, _i, _p]),
    "st_timestep_features": (_i, [_p, _l, _p, _p, _i, _i, _i, _p, _i, _p]),
    "st_timestep_sincos": (_i, [_p, _p, _p, _l, _i, _p]),
    "st_quantize_fp8": (_i, [_p, _l, _p, _p, _i, _i, _i, _p]),
    "st_layer_norm_quantize_fp8": (_i, [_p, _p, _p, _p, _p, _i, _i, _f, _i, _p]),
    "st_linear_fp8": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p, _z, _p, _z, _p]),
    "st_linear_emit8": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _p, _z, _p, _i, _p, _p, _i, _p, _p, _l, _p, _p, _p, _z, _p]),
    "st_linear_fp8x": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p, _i, _p, _p, _f, _p, _i, _p, _p, _l, _p, _p, _p, _z, _p, _z, _p]),
    "st_fp8_update_scales": (_i, [_p, _p, _p, _i, _f, _p]),
    "st_split_f32": (_i, [_p, _p, _l, _i, _l, _p]),
    "st_arm_split_output": (_i, [_p, _l, _i]),
    "st_attention_split": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _l, _l, _f, _p]),
}

_lib = None


class BackendError(RuntimeError):
    pass


def load(path=None):
    """Load (once) and return the operator library; raise if it is not built.  `path`: developer tools (tools/*.py A/B runs
    on -D variant builds) name another build of the same ABI explicitly, before anything else has loaded the product one."""
    global _lib
    if _lib is not None:
        if path is not None and os.path.abspath(path) != _lib._name:
            raise BackendError(f"operator library already loaded from {_lib._name}")
        return _lib
    path = os.path.abspath(path) if path is not None else lib_path()
    if not os.path.exists(path):
        raise BackendError(
            f"{path} not found: the HIP operator library is not built. "
            "Run `python -m stabletriton_amd.build` (needs hipcc); there is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    got = lib.st_abi_version()
    if got != ABI_VERSION:
        raise BackendError(f"ABI mismatch: library {got}, binding {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        lib = load()
        msg = lib.st_last_error().decode(errors="replace")
        lib.st_arm_split_output(None, 0, 0)      # a rejected launch may not have reached the arm: no later launch may write the image the caller is about to drop
        raise BackendError(f"{what}: {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return ST_BF16
    if dt == torch.float16:
        return ST_F16
    if dt == torch.float32:
        return ST_F32
    raise BackendError(f"unsupported dtype {dt}: the HIP operators take bfloat16, float16 or float32")


def stream_ptr() -> int:
    """Current torch HIP stream (launches must follow torch's stream so they
    are captured by hipGraph capture; reference: optimizers/cuda/graphs.py:72-108)."""
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors) -> None:
    for t in tensors:
        if t is not None and t.device.type != "cuda":
            raise BackendError("HIP operator called with a non-GPU tensor; there is no CPU fallback")
