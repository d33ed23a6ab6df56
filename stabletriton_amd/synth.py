"""Deterministic synthetic weights and inputs (no network, no checkpoints).

Every value is a pure function of (seed, tensor name, element index) computed
with 32-bit integer hashing carried in int64 lanes, so CPU, GPU and any torch
version produce bit-identical tensors.  The reference loads real SDXL weights
from the hub (implementations/Diffusers/load_sdxl_pipeline.py:17-25); the
state_dict schema it relies on (unet_pt.py:416-467) is the only contract, so
synthetic tensors are keyed by those same parameter names.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch

_M32 = 0xFFFFFFFF


def _mix32(x: torch.Tensor) -> torch.Tensor:
    # lowbias32-style finaliser on uint32 values held in int64 lanes.
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _key(seed: int, name: str) -> int:
    return (zlib.crc32(name.encode()) ^ ((seed * 0x9E3779B1) & _M32)) & _M32


_CHUNK = 1 << 20        # hash in cache-sized pieces; the values do not depend on the chunking


def uniform_pm1(name: str, numel: int, seed: int, device="cpu", offset: int = 0) -> torch.Tensor:
    """fp32 uniform in [-1, 1), exact multiples of 2**-23."""
    key = _key(seed, name)
    out = torch.empty(numel, dtype=torch.float32, device=device)
    chunk = _CHUNK if torch.device(device).type == "cpu" else 1 << 26
    for lo in range(0, numel, chunk):
        n = min(chunk, numel - lo)
        idx = torch.arange(offset + lo, offset + lo + n, dtype=torch.int64, device=device)
        h = _mix32(((idx * 0x9E3779B1) & _M32) ^ key)
        h = _mix32(h + (idx >> 32))
        out[lo:lo + n] = (h >> 8).to(torch.float32) * (2.0 ** -23) - 1.0
    return out


def normal(name: str, shape: Iterable[int], seed: int, device="cpu") -> torch.Tensor:
    """fp32 approximately-normal values (sum of 4 uniforms, unit variance).

    Only exactly-rounded fp32 adds/multiplies are used, so results are
    bit-identical across devices.
    """
    shape = tuple(shape)
    n = int(math.prod(shape)) if shape else 1
    acc = torch.zeros(n, dtype=torch.float32, device=device)
    for j in range(4):
        acc = acc + uniform_pm1(f"{name}#{j}", n, seed, device)
    # var(sum of 4 U[-1,1)) = 4/3
    return (acc * 0.8660254037844386).reshape(shape)


def param_tensor(name: str, shape: Tuple[int, ...], seed: int, device="cpu") -> torch.Tensor:
    """Synthetic value for one named parameter (fp32).

    >=2-D (Linear / Conv weights): U[-1,1) * sqrt(3 / fan_in) -> unit gain.
    1-D '.weight' (norm scales): 1 + 0.1 U.   1-D '.bias': 0.05 U.
    """
    shape = tuple(shape)
    n = int(math.prod(shape))
    u = uniform_pm1(name, n, seed, device)
    if len(shape) >= 2:
        fan_in = int(math.prod(shape[1:]))
        u = u * float(math.sqrt(3.0 / fan_in))
    elif name.endswith(".weight"):
        u = u * 0.1 + 1.0
    else:
        u = u * 0.05
    return u.reshape(shape)


@torch.no_grad()
def fill_module_(module: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    """Overwrite every parameter of `module` in place (on its own device/dtype)."""
    for name, p in module.named_parameters():
        p.copy_(param_tensor(name, tuple(p.shape), seed, p.device).to(p.dtype))
    return module


def state_dict_for(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    return {k: param_tensor(k, s, seed, device) for k, s in shapes.items()}


def latent_size(latent_hw) -> Tuple[int, int]:
    """(height, width) of a latent given as one side (square) or as a pair (SDXL's aspect buckets: 152 x 104 ...)."""
    if isinstance(latent_hw, (tuple, list)):
        h, w = latent_hw
        return int(h), int(w)
    return int(latent_hw), int(latent_hw)


def denoise_inputs(batch: int, latent_hw, seed: int = 1234, device="cpu",
                   cross_dim: int = 2048, pooled_dim: int = 1280, tokens: int = 77, n_time_ids: int = 6):
    """SURVEY.md 8(d) synthetic inputs: unit-normal latent (caller scales by
    init sigma), text states, pooled text embedding and SDXL time ids.  `latent_hw`: one side, or (height, width)."""
    lh, lw = latent_size(latent_hw)
    ph, pw = float(lh * 8), float(lw * 8)
    return {
        "latent": normal("latent", (batch, 4, lh, lw), seed, device),
        "encoder_hidden_states": normal("ehs", (batch, tokens, cross_dim), seed, device),
        "text_embeds": normal("text_embeds", (batch, pooled_dim), seed, device),
        # base: (original h, w, crop top, left, target h, w); refiner (5 ids): (original h, w, crop top, left, aesthetic score)
        "time_ids": torch.tensor([[ph, pw, 0.0, 0.0, ph, pw][:n_time_ids] if n_time_ids >= 6 else [ph, pw, 0.0, 0.0, 6.0][:n_time_ids]] * batch,
                                 dtype=torch.float32, device=device),
    }
