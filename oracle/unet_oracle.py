"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

CPU fp32 restatement of the reference's eager SDXL UNet forward
(/root/reference/src/stabletriton/optimizers/unet_pt.py), written as plain
functions over a Diffusers-keyed state_dict.  Only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may import this package; the
product path (stabletriton_amd/) never does.

Pinning: oracle/make_golden.py imports the reference module in the build
container, loads identical synthetic weights into it and records its outputs
under tests/golden/; tests/test_oracle_golden.py checks this restatement
against those vectors.  The reference has no golden vectors of its own
(SURVEY.md section 4).

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# ---- storage emulation (tests only): what 16-bit STORAGE alone costs --------------------------------------------------
# `with storage(torch.bfloat16):` rounds the result of every operator below (and, through `rounded_state_dict`, every weight)
# to that type and back to fp32, the arithmetic inside an operator staying fp32: the deviation of such a run from the fp32
# run is the error that 16-bit tensors at operator boundaries cause by themselves, whatever the kernels do.  The bf16 / fp16
# gates of tests/test_unet_gpu.py are a small factor of THAT (oracle/make_rounded_golden.py records it), not of a measurement
# of the kernels under test.  The compiled graph rounds in fewer places (LayerNorm, GEGLU and the residual adds live inside
# GEMM epilogues), so the emulation is an upper-side estimate of the storage error, not a model of the kernels.
_STORAGE = None


class storage:
    def __init__(self, dtype: Optional[torch.dtype]):
        self.dtype = dtype

    def __enter__(self):
        global _STORAGE
        self.prev, _STORAGE = _STORAGE, self.dtype
        return self

    def __exit__(self, *exc):
        global _STORAGE
        _STORAGE = self.prev
        return False


def _r(x: torch.Tensor) -> torch.Tensor:
    return x if _STORAGE is None else x.to(_STORAGE).float()


def rounded_state_dict(sd: SD, dtype: torch.dtype) -> SD:
    return {k: v.to(dtype).float() for k, v in sd.items()}


# ---- fp8-plan emulation (tests only): what e4m3 OPERANDS on the planned projections cost -----------------------------------
# `with fp8_plan():` (inside `storage(torch.bfloat16)`: the fp8 mode runs on a bf16 model) multiplies the projections of the fp8
# plan (stabletriton_amd/optimizers/plan_fp8.py: the LayerNorm-fed q|k|v, attn2.to_q and GEGLU projections and the feed-forward
# output projection of every transformer block) the way the product does, in fp32 arithmetic: the activation as ONE e4m3 tensor
# under a per-tensor scale of margin x max|x| / 448 (the delayed scale: the previous evaluation's maximum - on a single step the
# calibration pass has seen this very tensor), the weight as e4m3 under per-output-channel scales, the LayerNorm folded around the
# product (row statistics of the unquantised values; gamma folded into the weight BEFORE it is quantised; the GEGLU output kept
# only as e4m3).  The deviation of such a run from the reference is what the operand format costs by itself; the gate of
# tests/test_fp8_gpu.py is a small factor of that (oracle/make_rounded_golden.py f1_fp8), not of a measurement of the kernels.
_FP8_PLAN = False
FP8_MAX, FP8_MARGIN = 448.0, 2.0          # (stabletriton_amd/ops.py: FP8_MAX, FP8_MARGIN)


class fp8_plan:
    def __enter__(self):
        global _FP8_PLAN
        self.prev, _FP8_PLAN = _FP8_PLAN, True
        return self

    def __exit__(self, *exc):
        global _FP8_PLAN
        _FP8_PLAN = self.prev
        return False


def _e4m3_act(x: torch.Tensor) -> torch.Tensor:
    s = x.abs().amax().clamp_min(1e-12) * FP8_MARGIN / FP8_MAX
    return (x / s).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float() * s


def _e4m3_weight(w: torch.Tensor) -> torch.Tensor:
    s = w.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / FP8_MAX
    return (w / s).to(torch.float8_e4m3fn).float() * s


def ln_linear_fp8(sd: SD, ln_p: str, lin_ps, x):
    """LayerNorm(x) -> the projections `lin_ps` as the fp8 plan computes them (st_linear_fp8x with a folded LayerNorm)."""
    g, b = sd[ln_p + ".weight"], sd[ln_p + ".bias"]
    mean = x.mean(-1, keepdim=True)
    rstd = (x.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    xn = (_e4m3_act(x) - mean) * rstd                  # the e4m3 copy of the un-normalised tensor, the statistics of the stored values
    outs = []
    for lp in lin_ps:
        w = sd[lp + ".weight"]
        d = w @ b
        if lp + ".bias" in sd:
            d = d + sd[lp + ".bias"]
        outs.append(_r(xn @ _e4m3_weight(w * g[None, :]).T + d))
    return outs


def linear_fp8(sd: SD, p: str, x8):
    """x8 (already e4m3 values) times the e4m3 weight (+ bias): the feed-forward output projection of the plan."""
    y = x8 @ _e4m3_weight(sd[p + ".weight"]).T
    return y + sd[p + ".bias"] if p + ".bias" in sd else y


def _has(sd: SD, key: str) -> bool:
    return key in sd


def _count(sd: SD, fmt: str) -> int:
    n = 0
    while fmt.format(n) in sd:
        n += 1
    return n


def linear(sd: SD, p: str, x):
    return _r(F.linear(x, sd[p + ".weight"], sd.get(p + ".bias")))


def conv(sd: SD, p: str, x, stride=1, padding=1):
    return _r(F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding))


def group_norm(sd: SD, p: str, x, eps: float, groups: int = 32):
    return _r(F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps))


def layer_norm(sd: SD, p: str, x):
    w = sd[p + ".weight"]
    return _r(F.layer_norm(x, w.shape, w, sd[p + ".bias"], 1e-5))


def timestep_features(t: torch.Tensor, dim: int) -> torch.Tensor:
    """unet_pt.py:22-36 - cos first, then sin; exponent = -ln(1e4) * i / half."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32)
    exponent = exponent / (half - 0.0)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def timestep_mlp(sd: SD, p: str, x):
    """unet_pt.py:46-51."""
    return linear(sd, p + ".linear_2", _r(F.silu(linear(sd, p + ".linear_1", x))))


def resnet_block(sd: SD, p: str, x, temb, groups: int = 32):
    """unet_pt.py:74-95."""
    h = _r(F.silu(group_norm(sd, p + ".norm1", x, 1e-5, groups)))
    h = conv(sd, p + ".conv1", h)
    h = _r(h + linear(sd, p + ".time_emb_proj", _r(F.silu(temb)))[:, :, None, None])
    h = _r(F.silu(group_norm(sd, p + ".norm2", h, 1e-5, groups)))
    h = conv(sd, p + ".conv2", h)
    if _has(sd, p + ".conv_shortcut.weight"):
        x = conv(sd, p + ".conv_shortcut", x, padding=0)
    return _r(x + h)


def attention_core(q, k, v, heads: int):
    """unet_pt.py:133-142: softmax(q k^T / sqrt(d)) v per head; q,k,v are (B,T,H*d)."""
    b, t, c = q.shape
    d = c // heads
    q = q.view(b, t, heads, d).transpose(1, 2)
    k = k.view(b, k.shape[1], heads, d).transpose(1, 2)
    v = v.view(b, v.shape[1], heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-2, -1)) * (d ** -0.5)
    o = torch.matmul(torch.softmax(s, dim=-1), v)
    return _r(o.transpose(1, 2).contiguous().view(b, t, c))


def attention(sd: SD, p: str, x, context=None, head_dim: int = 64):
    """unet_pt.py:121-147."""
    src = x if context is None else context
    q = linear(sd, p + ".to_q", x)
    k = linear(sd, p + ".to_k", src)
    v = linear(sd, p + ".to_v", src)
    o = attention_core(q, k, v, q.shape[-1] // head_dim)
    return linear(sd, p + ".to_out.0", o)


def geglu(x_proj):
    """unet_pt.py:155-158: exact-erf GELU on the second half."""
    a, g = x_proj.chunk(2, dim=-1)
    return _r(a * F.gelu(g))


def _transformer_layer_fp8(sd: SD, p: str, x, context, head_dim: int):
    """transformer_layer with the planned projections on e4m3 operands (see fp8_plan); everything else as below."""
    q, k, v = ln_linear_fp8(sd, p + ".norm1", [p + ".attn1.to_q", p + ".attn1.to_k", p + ".attn1.to_v"], x)
    x = _r(linear(sd, p + ".attn1.to_out.0", attention_core(q, k, v, q.shape[-1] // head_dim)) + x)
    (q,) = ln_linear_fp8(sd, p + ".norm2", [p + ".attn2.to_q"], x)
    k, v = linear(sd, p + ".attn2.to_k", context), linear(sd, p + ".attn2.to_v", context)
    x = _r(linear(sd, p + ".attn2.to_out.0", attention_core(q, k, v, q.shape[-1] // head_dim)) + x)
    (hp,) = ln_linear_fp8(sd, p + ".norm3", [p + ".ff.net.0.proj"], x)
    a, gte = hp.chunk(2, dim=-1)
    h8 = _e4m3_act(a * F.gelu(gte))                     # the GEGLU projection leaves ONLY the e4m3 copy of its output
    return _r(linear_fp8(sd, p + ".ff.net.2", h8) + x)


def transformer_layer(sd: SD, p: str, x, context, head_dim: int = 64):
    """unet_pt.py:189-210."""
    if _FP8_PLAN:
        return _transformer_layer_fp8(sd, p, x, context, head_dim)
    x = _r(attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", x), None, head_dim) + x)
    x = _r(attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", x), context, head_dim) + x)
    h = geglu(linear(sd, p + ".ff.net.0.proj", layer_norm(sd, p + ".norm3", x)))
    return _r(linear(sd, p + ".ff.net.2", h) + x)


def spatial_transformer(sd: SD, p: str, x, context, groups: int = 32, head_dim: int = 64):
    """unet_pt.py:223-243 (GroupNorm eps is 1e-6 here, unet_pt.py:216)."""
    b, c, h, w = x.shape
    y = group_norm(sd, p + ".norm", x, 1e-6, groups)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = linear(sd, p + ".proj_in", y)
    for i in range(_count(sd, p + ".transformer_blocks.{}.norm1.weight")):
        y = transformer_layer(sd, f"{p}.transformer_blocks.{i}", y, context, head_dim)
    y = linear(sd, p + ".proj_out", y)
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
    return _r(y + x)


def unet_forward(sd: SD, sample, timestep, encoder_hidden_states, text_embeds, time_ids,
                 groups: int = 32, head_dim: int = 64, time_proj_dim: Optional[int] = None,
                 add_time_proj_dim: Optional[int] = None) -> torch.Tensor:
    """unet_pt.py:469-542.  Topology is read off the state_dict keys."""
    if time_proj_dim is None:
        time_proj_dim = sd["time_embedding.linear_1.weight"].shape[1]
    if add_time_proj_dim is None:
        add_time_proj_dim = (sd["add_embedding.linear_1.weight"].shape[1] - text_embeds.shape[1]) // time_ids.shape[1]
    b = sample.shape[0]
    t = timestep.reshape(-1).expand(b)
    emb = timestep_mlp(sd, "time_embedding", timestep_features(t, time_proj_dim).to(sample.dtype))
    tid = timestep_features(time_ids.flatten(), add_time_proj_dim).reshape(b, -1)
    add = torch.cat([text_embeds, tid], dim=-1).to(emb.dtype)
    emb = _r(emb + timestep_mlp(sd, "add_embedding", add))

    x = conv(sd, "conv_in", sample)
    skips: List[torch.Tensor] = [x]
    n_down = _count(sd, "down_blocks.{}.resnets.0.norm1.weight")
    for i in range(n_down):
        p = f"down_blocks.{i}"
        for j in range(_count(sd, p + ".resnets.{}.norm1.weight")):
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, groups)
            if _has(sd, f"{p}.attentions.{j}.norm.weight"):
                x = spatial_transformer(sd, f"{p}.attentions.{j}", x, encoder_hidden_states, groups, head_dim)
            skips.append(x)
        if _has(sd, p + ".downsamplers.0.conv.weight"):
            x = conv(sd, p + ".downsamplers.0.conv", x, stride=2)      # unet_pt.py:246-254
            skips.append(x)

    x = resnet_block(sd, "mid_block.resnets.0", x, emb, groups)        # unet_pt.py:404-413
    x = spatial_transformer(sd, "mid_block.attentions.0", x, encoder_hidden_states, groups, head_dim)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, groups)

    for i in range(_count(sd, "up_blocks.{}.resnets.0.norm1.weight")):
        p = f"up_blocks.{i}"
        for j in range(_count(sd, p + ".resnets.{}.norm1.weight")):
            x = torch.cat([x, skips.pop()], dim=1)                     # unet_pt.py:352-357
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, groups)
            if _has(sd, f"{p}.attentions.{j}.norm.weight"):
                x = spatial_transformer(sd, f"{p}.attentions.{j}", x, encoder_hidden_states, groups, head_dim)
        if _has(sd, p + ".upsamplers.0.conv.weight"):
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")     # unet_pt.py:264-266
            x = conv(sd, p + ".upsamplers.0.conv", x)

    x = _r(F.silu(group_norm(sd, "conv_norm_out", x, 1e-5, groups)))   # unet_pt.py:538-540
    return conv(sd, "conv_out", x)


def euler_img2img(unet_fn, init_latent, noise_unit, tables, strength: float) -> torch.Tensor:
    """img2img form (restated diffusers img2img `get_timesteps` + `add_noise`; third-party, parity unpinned): start at
    schedule entry t_start = n - int(n * strength) from init + noise * sigma[t_start]."""
    n = tables.n_steps
    t_start = max(n - min(int(n * strength), n), 0)
    x = init_latent.float() + noise_unit.float() * float(tables.sigmas[t_start])
    in_scale, dsigma = tables.in_scale(), tables.dsigma()
    for i in range(t_start, n):
        eps = unet_fn(x * float(in_scale[i]), torch.tensor(float(tables.timesteps[i])))
        x = x + eps.float() * float(dsigma[i])
    return x


def euler_denoise(unet_fn, latent_unit, tables, n_steps: Optional[int] = None, state_dtype=torch.float32) -> torch.Tensor:
    """Euler-discrete epsilon-prediction loop (restated diffusers==0.21.2
    EulerDiscreteScheduler, see stabletriton_amd/scheduler.py header; parity of
    the scheduler arithmetic itself is unpinned).  `unet_fn(x_in, t)` returns
    eps; `latent_unit` is unit-variance noise; state is kept in fp32 (`state_dtype=torch.float64`: the
    same loop with the same fp32 table constants in double precision - oracle/make_golden.py's truth runs)."""
    x = latent_unit.to(state_dtype) * tables.init_noise_sigma
    n = tables.n_steps if n_steps is None else n_steps
    in_scale, dsigma = tables.in_scale(), tables.dsigma()
    for i in range(n):
        x_in = x * float(in_scale[i])
        eps = unet_fn(x_in, torch.tensor(float(tables.timesteps[i])))
        x = x + eps.to(state_dtype) * float(dsigma[i])
    return x


def euler_denoise_cfg(unet_fn, latent_unit, tables, guidance_scale: float, n_steps: Optional[int] = None,
                      state_dtype=torch.float32) -> torch.Tensor:
    """Classifier-free-guidance form of the loop, as the reference's call site runs it
    (implementations/Diffusers/load_sdxl_pipeline.py:39-46 -> diffusers 0.21.2
    StableDiffusionXLPipeline.__call__: `latent_model_input = cat([latents] * 2)`,
    `noise_pred = uncond + g * (text - uncond)`; third-party arithmetic, parity unpinned like
    the scheduler).  `unet_fn(x_in2, t)` takes the duplicated (2, ...) input, row 0 = negative
    conditioning; `latent_unit` is (1, ...)."""
    x = latent_unit.to(state_dtype) * tables.init_noise_sigma
    n = tables.n_steps if n_steps is None else n_steps
    in_scale, dsigma = tables.in_scale(), tables.dsigma()
    for i in range(n):
        x_in = torch.cat([x, x]) * float(in_scale[i])
        eps2 = unet_fn(x_in, torch.tensor(float(tables.timesteps[i]))).to(state_dtype)
        eps = eps2[0:1] + guidance_scale * (eps2[1:2] - eps2[0:1])
        x = x + eps * float(dsigma[i])
    return x
