"""Euler-discrete sampling tables for the denoise loop.

The reference does not contain a scheduler: its 50-step loop is the
third-party Diffusers SDXL pipeline (diffusers==0.21.2, requirements.txt:1;
call site implementations/Diffusers/load_sdxl_pipeline.py:39-46) driving
`pipe.unet` once per step.  This module restates the published
EulerDiscreteScheduler arithmetic with the SDXL-base scheduler settings
(scaled-linear betas 0.00085..0.012, 1000 train steps, "leading" spacing,
steps_offset 1, epsilon prediction) so the loop can run on-device inside a
hipGraph.  Parity for this arithmetic is not pinned by any reference test
(SURVEY.md 8c-ii): the oracle drives the reference UNet with the same tables.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class EulerTables:
    timesteps: np.ndarray      # (n,) float32, value fed to the UNet each step
    sigmas: np.ndarray         # (n+1,) float32, last entry 0
    init_noise_sigma: float

    @property
    def n_steps(self) -> int:
        return len(self.timesteps)

    def in_scale(self) -> np.ndarray:
        """1/sqrt(sigma^2+1): latent -> UNet input scaling per step."""
        s = self.sigmas[:-1].astype(np.float64)
        return (1.0 / np.sqrt(s * s + 1.0)).astype(np.float32)

    def dsigma(self) -> np.ndarray:
        """sigma[i+1]-sigma[i]: x <- x + eps * dsigma (epsilon prediction)."""
        s = self.sigmas.astype(np.float64)
        return (s[1:] - s[:-1]).astype(np.float32)


def euler_discrete_tables(n_steps: int = 50, n_train: int = 1000, beta_start: float = 0.00085,
                          beta_end: float = 0.012, steps_offset: int = 1) -> EulerTables:
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n_train, dtype=np.float64) ** 2
    alphas_cumprod = np.cumprod(1.0 - betas)
    all_sigmas = np.sqrt((1.0 - alphas_cumprod) / alphas_cumprod)
    ratio = n_train // n_steps
    ts = (np.arange(0, n_steps) * ratio).round()[::-1].astype(np.float64) + steps_offset
    sig = np.interp(ts, np.arange(n_train, dtype=np.float64), all_sigmas)
    sig = np.concatenate([sig, [0.0]])
    init = float(np.sqrt(sig.max() ** 2 + 1.0))
    return EulerTables(ts.astype(np.float32), sig.astype(np.float32), init)
