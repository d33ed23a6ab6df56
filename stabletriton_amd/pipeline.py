"""The 50-step denoise loop on the device, captured as a hipGraph.

The reference captures ONE UNet forward per graph and leaves the loop to the
Diffusers pipeline (optimizers/cuda/graphs.py:100-110,
implementations/Diffusers/load_sdxl_pipeline.py:39-46).  BASELINE.json's
north_star asks for the whole loop as a hipGraph: here every step is
[UNet forward -> Euler update of the fp32 latent -> scaled bf16 input of the next
step], all on the device, so the 50 steps can be captured back to back and a
run is one graph launch.  `mode="step"` captures a single step driven by a
device-side step counter instead (replayed n times), `mode="eager"` captures
nothing (used for per-kernel timing and debugging).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import ops
from .scheduler import EulerTables, euler_discrete_tables


class DenoiseLoop:
    def __init__(self, unet: Callable, batch: int, latent_hw, dtype: torch.dtype, device,
                 tables: Optional[EulerTables] = None, cross_dim: int = 2048, pooled_dim: int = 1280,
                 tokens: int = 77, mode: str = "loop", n_time_ids: int = 6):
        assert mode in ("loop", "step", "eager")
        self.unet, self.mode, self.dtype = unet, mode, dtype
        self.device = torch.device(device)
        self.tables = tables or euler_discrete_tables(50)
        n = self.n_steps = self.tables.n_steps
        dev = self.device
        cl = torch.channels_last
        # `latent_hw`: one side of a square latent, or (height, width) - SDXL's aspect buckets (1216 x 832 px = 152 x 104);
        # both sides multiples of 4: the UNet halves the latent twice (Downsample2D, unet_pt.py:246-256) and doubles it back
        lh, lw = (int(latent_hw[0]), int(latent_hw[1])) if isinstance(latent_hw, (tuple, list)) else (int(latent_hw), int(latent_hw))
        self.latent = torch.zeros((batch, 4, lh, lw), dtype=torch.float32, device=dev).contiguous(memory_format=cl)
        self.x_in = torch.zeros((batch, 4, lh, lw), dtype=dtype, device=dev).contiguous(memory_format=cl)
        self.ehs = torch.zeros((batch, tokens, cross_dim), dtype=dtype, device=dev)
        self.text_embeds = torch.zeros((batch, pooled_dim), dtype=dtype, device=dev)
        self.time_ids = torch.zeros((batch, n_time_ids), dtype=dtype, device=dev)
        self.timesteps = torch.tensor(self.tables.timesteps, dtype=torch.float32, device=dev)
        self.dsigma = torch.tensor(self.tables.dsigma(), dtype=torch.float32, device=dev)
        self.in_scale = torch.tensor(self.tables.in_scale(), dtype=torch.float32, device=dev)
        self.step_ids = torch.arange(n, dtype=torch.int32, device=dev)      # constants for the unrolled loop
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)           # counter for mode="step"
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._captured_steps = 0
        # text-context projections are step-invariant: evaluated once per prompt when the compiled
        # UNet exposes the split (optimization._install_context_split)
        self._split = hasattr(unet, "precompute_context") and hasattr(unet, "forward_with_context")
        self.ctx = None
        # the time path (timestep features, embedding MLPs, resnet time projections) depends only on the
        # schedule entry and the added conditioning: one table row per step, filled in set_conditioning
        self._tsplit = self._split and hasattr(unet, "precompute_time")
        self.time_tables = None

    # ---- inputs --------------------------------------------------------------------------
    def set_conditioning(self, encoder_hidden_states, text_embeds, time_ids) -> None:
        self.ehs.copy_(encoder_hidden_states)
        self.text_embeds.copy_(text_embeds)
        self.time_ids.copy_(time_ids)
        self.refresh_weights()
        if self._split:
            with torch.no_grad():
                new = self.unet.precompute_context(self.ehs)
            if self.ctx is None:
                self.ctx = tuple(t.clone() for t in new)          # static buffers the captured graph reads
            else:
                for dst, src in zip(self.ctx, new):
                    dst.copy_(src)
        if self._tsplit:
            # every schedule entry in ONE pass: the time path sees n_steps * batch rows (row s*B + b = step s, sample b),
            # so its GEMMs run once with M = n_steps * batch instead of n_steps times with M = batch
            n, b = self.n_steps, self.ehs.shape[0]
            with torch.no_grad():
                rows = self.x_in.new_empty((n * b, 1, 1, 1))                       # read for its batch size and dtype only
                cond = {"text_embeds": self.text_embeds.repeat(n, 1), "time_ids": self.time_ids.repeat(n, 1)}
                out = self.unet.precompute_time(rows, self.timesteps.repeat_interleave(b), cond)
                new = tuple(o.reshape(n, b, *o.shape[1:]) for o in out)
            if self.time_tables is None:
                self.time_tables = tuple(t.clone() for t in new)
            else:
                for dst, src in zip(self.time_tables, new):
                    dst.copy_(src)

    def refresh_weights(self) -> int:
        """Captured graphs read derived weight buffers (fused q|k|v, LayerNorm-folded projections) by address: after an
        in-place weight update (LoRA merge) re-derive them in place.  Called per prompt; call it yourself after updating
        weights between two runs of the same prompt."""
        ectx = getattr(self.unet, "exec_context", None)
        return ectx.refresh_derived(full=True) if ectx is not None else 0

    def set_noise(self, latent_unit: torch.Tensor) -> None:
        """latent_unit ~ N(0,1); scaled by the scheduler's init sigma (fp32 state)."""
        self.latent.copy_(latent_unit.to(self.device, torch.float32) * self.tables.init_noise_sigma)
        self.x_in.copy_(self.latent * float(self.tables.in_scale()[0]))
        self.step.zero_()
        self._recalibrate(0)

    def set_image(self, init_latent: torch.Tensor, noise_unit: torch.Tensor, strength: float) -> int:
        """img2img start (the refiner's use, BASELINE config #5; restated diffusers img2img: `get_timesteps` +
        `scheduler.add_noise`): skip the first n - int(n * strength) schedule entries, start from
        init_latent + noise * sigma[t_start].  Returns the number of steps left to run (`run_steps(k)`, mode step / eager)."""
        if self.mode == "loop":
            raise ValueError("set_image needs mode='step' or 'eager': the captured full-trajectory loop cannot start mid-schedule")
        n = self.n_steps
        t_start = max(n - min(int(n * strength), n), 0)
        if t_start >= n:
            raise ValueError(f"strength {strength} leaves no denoise step of the {n}-step schedule (int(n * strength) == 0)")
        sigma = float(self.tables.sigmas[t_start])
        lat = init_latent.to(self.device, torch.float32) + noise_unit.to(self.device, torch.float32) * sigma
        self.latent.copy_(lat)
        self.x_in.copy_(self.latent * float(self.tables.in_scale()[t_start]))
        self.step.fill_(t_start)
        self._recalibrate(t_start)
        return n - t_start

    def _recalibrate(self, i: int) -> None:
        """fp8 plan only: a trajectory starts from scales measured on its own first evaluation (not on the last step of
        whatever ran before), so the same inputs always give the same outputs."""
        if self.ctx is None and self._split:
            return                                   # no prompt yet: the first evaluation after compile measures by itself
        from .optimization import recalibrate_fp8
        row = tuple(tbl[i] for tbl in self.time_tables) if (self._tsplit and self.time_tables is not None) else None
        recalibrate_fp8(self.unet, lambda: self._unet(self.timesteps[i], row))

    # ---- one step ------------------------------------------------------------------------
    def _cond(self) -> Dict[str, torch.Tensor]:
        return {"text_embeds": self.text_embeds, "time_ids": self.time_ids}

    def _unet(self, t, time_row=None):
        if self._split:
            if self.ctx is None:
                raise RuntimeError("set_conditioning() must be called before running the loop")
            if self._tsplit:
                return self.unet.forward_with_context(self.x_in, t, self.ctx, self._cond(), time_cache=time_row)[0]
            return self.unet.forward_with_context(self.x_in, t, self.ctx, self._cond())[0]
        return self.unet(self.x_in, t, self.ehs, self._cond())[0]

    def _step_const(self, i: int) -> None:
        row = tuple(tbl[i] for tbl in self.time_tables) if self._tsplit else None      # static views: no launch
        eps = self._unet(self.timesteps[i], row)
        ops.euler_step(self.latent, eps, self.x_in, self.dsigma, self.in_scale, self.step_ids[i:i + 1])

    def _step_counted(self) -> None:
        idx = self.step.long()
        t = self.timesteps.index_select(0, idx)[0]
        row = tuple(tbl.index_select(0, idx)[0] for tbl in self.time_tables) if self._tsplit else None
        eps = self._unet(t, row)
        ops.euler_step(self.latent, eps, self.x_in, self.dsigma, self.in_scale, self.step)
        ops.step_advance(self.step, self.n_steps)

    # ---- capture / run -------------------------------------------------------------------
    def capture(self, warmup: int = 1) -> None:
        if self.mode == "eager" or self.graph is not None:
            return
        keep = (self.latent.clone(), self.x_in.clone(), self.step.clone())
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step_counted() if self.mode == "step" else self._step_const(0)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.latent.copy_(keep[0]); self.x_in.copy_(keep[1]); self.step.copy_(keep[2])
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            if self.mode == "step":
                self._step_counted()
                self._captured_steps = 1
            else:
                for i in range(self.n_steps):
                    self._step_const(i)
                self._captured_steps = self.n_steps
        self.graph = g
        self.latent.copy_(keep[0]); self.x_in.copy_(keep[1]); self.step.copy_(keep[2])
        # (fp8 plan: the warm-up evaluation above measured the scales its own way; start them over exactly as set_noise /
        #  set_image do, so the first trajectory after a capture equals every later one)
        self._recalibrate(int(keep[2].item()) % self.n_steps)

    def run_steps(self, k: int) -> None:
        """Advance exactly k denoise steps from the current state (asynchronous)."""
        if self.mode == "eager":
            s = int(self.step.item())
            for i in range(k):
                self._step_const((s + i) % self.n_steps)
            self.step.fill_((s + k) % self.n_steps)
            return
        self.capture()
        if self.mode == "loop":
            if k % self.n_steps != 0:
                raise ValueError(f"mode='loop' runs whole {self.n_steps}-step loops; got k={k}")
            for _ in range(k // self.n_steps):
                self.graph.replay()
        else:
            for _ in range(k):
                self.graph.replay()

    def denoise(self, latent_unit: torch.Tensor) -> torch.Tensor:
        """Full trajectory: unit noise in, final fp32 latent (NCHW contiguous) out."""
        self.set_noise(latent_unit)
        self.run_steps(self.n_steps)
        return self.latent.contiguous(memory_format=torch.contiguous_format).clone()
