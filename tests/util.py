"""Shared helpers for the parity tests (oracle = checker only)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerances, written down once.
#  fp32 ("strict") path: same arithmetic as the eager reference up to summation
#  order -> 2e-4 of the output scale per op, 1e-3 abs on final latents (north_star).
#  bf16 path: inputs/outputs carry 8 mantissa bits (rel 2^-9 = 2e-3 per rounding).
#  fp16 path (the reference's own compute type): 11 mantissa bits (rel 2^-12 per rounding): a quarter of the bf16 bound.
TOL = {torch.float32: 2e-4, torch.bfloat16: 2e-2, torch.float16: 5e-3}
HALF_DTYPES = [torch.bfloat16, torch.float16]


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(out: torch.Tensor, ref: torch.Tensor) -> float:
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    return float((out - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def assert_close(out, ref, dtype, what="", factor=1.0):
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out.float()).all(), f"{what}: non-finite output"
    e = rel_err(out, ref)
    assert e <= TOL[dtype] * factor, f"{what}: max err / max|ref| = {e:.3e} > {TOL[dtype] * factor:.1e}"


def rounded(x: torch.Tensor, dtype) -> torch.Tensor:
    """Value the kernel actually sees (bf16- / fp16-rounded), as fp32 for the oracle."""
    return x.to(dtype).float()
