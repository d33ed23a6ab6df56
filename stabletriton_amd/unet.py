"""SDXL-family conditional UNet as a plain eager nn.Module: the "model under
compile" handed to `optimize_model` (role of the reference's
src/stabletriton/optimizers/unet_pt.py:416-542).

Parameter names follow the Diffusers UNet2DConditionModel state_dict so real
SDXL checkpoints load with `load_state_dict` (reference:
implementations/Diffusers/load_sdxl_pipeline.py:24-25).  The network is built
from a `UNetSpec` rather than hard-coded stages, so the same code gives the
SDXL-base network, and the small networks the tests run.

The eager ops are written in the canonical form the rewrite passes look for
(see stabletriton_amd/passes): attention is
view->transpose->matmul*scale->softmax->matmul->transpose->contiguous->view,
GEGLU is chunk -> a * gelu(b).
"""
from __future__ import annotations

from collections import namedtuple
from dataclasses import dataclass
import math
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn
import torch.nn.functional as F


@dataclass(frozen=True)
class UNetSpec:
    in_channels: int = 4
    out_channels: int = 4
    widths: Tuple[int, ...] = (320, 640, 1280)        # channels per resolution level
    depths: Tuple[int, ...] = (0, 2, 10)              # transformer layers per level (0 = no attention)
    resnets_per_level: int = 2
    head_dim: int = 64
    cross_dim: int = 2048
    temb_dim: int = 1280
    time_proj_dim: int = 320
    add_time_proj_dim: int = 256
    pooled_dim: int = 1280
    n_time_ids: int = 6
    groups: int = 32
    sample_size: int = 128
    mid_depth: Optional[int] = None                   # transformer layers of the middle block (None: those of the last level)

    @property
    def add_in_dim(self) -> int:
        return self.pooled_dim + self.n_time_ids * self.add_time_proj_dim


SDXL_BASE = UNetSpec()
# SDXL-refiner (BASELINE config #5).  The reference has no refiner model: the topology below is the
# published stabilityai/stable-diffusion-xl-refiner-1.0 UNet configuration (four levels 384/768/1536/1536, four
# transformer layers on the two middle levels and in the middle block, 1280-wide text context, five size / crop /
# aesthetic-score ids), restated by the oracle from the state_dict keys - parity unpinned.
SDXL_REFINER = UNetSpec(widths=(384, 768, 1536, 1536), depths=(0, 4, 4, 0), mid_depth=4, cross_dim=1280, temb_dim=1536,
                        time_proj_dim=384, add_time_proj_dim=256, pooled_dim=1280, n_time_ids=5)
# small network with the same topology, for CPU tests and quick GPU parity runs
TINY = UNetSpec(widths=(64, 128, 256), depths=(0, 1, 2), cross_dim=128, temb_dim=256,
                time_proj_dim=64, add_time_proj_dim=32, pooled_dim=64, sample_size=16)
TINY_REFINER = UNetSpec(widths=(64, 128, 256, 256), depths=(0, 1, 1, 0), mid_depth=1, cross_dim=128, temb_dim=256,
                        time_proj_dim=64, add_time_proj_dim=32, pooled_dim=64, n_time_ids=5, sample_size=16)


class SinusoidalProj(nn.Module):
    """cos|sin embedding of a 1-D tensor of positions (reference unet_pt.py:17-36)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        freqs = torch.exp(
            torch.arange(half, dtype=torch.float32, device=t.device) * (-math.log(10000.0)) / half
        )
        ang = t[:, None].float() * freqs[None, :]
        return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class TimestepMLP(nn.Module):
    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.linear_1 = nn.Linear(d_in, d_out)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(d_out, d_out)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class ResBlock(nn.Module):
    """GN-SiLU-conv3x3 (+temb) GN-SiLU-conv3x3 + skip (reference unet_pt.py:54-95)."""

    def __init__(self, c_in: int, c_out: int, temb_dim: int, groups: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, c_in, eps=1e-5)
        self.conv1 = nn.Conv2d(c_in, c_out, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, c_out)
        self.norm2 = nn.GroupNorm(groups, c_out, eps=1e-5)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(c_out, c_out, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(c_in, c_out, 1) if c_in != c_out else None

    def forward(self, x, temb):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """Multi-head attention, self or cross (reference unet_pt.py:98-147)."""

    def __init__(self, dim: int, head_dim: int, kv_dim: Optional[int] = None):
        super().__init__()
        self.num_heads = dim // head_dim
        self.head_dim = head_dim
        self.scale = head_dim ** -0.5
        kv_dim = dim if kv_dim is None else kv_dim
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(kv_dim, dim, bias=False)
        self.to_v = nn.Linear(kv_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])

    def forward(self, x, context=None):
        src = x if context is None else context
        q = self.to_q(x)
        k = self.to_k(src)
        v = self.to_v(src)
        b, t, c = q.size()
        q = q.view(q.size(0), q.size(1), self.num_heads, self.head_dim).transpose(1, 2)
        k = k.view(k.size(0), k.size(1), self.num_heads, self.head_dim).transpose(1, 2)
        v = v.view(v.size(0), v.size(1), self.num_heads, self.head_dim).transpose(1, 2)
        w = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) * self.scale, dim=-1)
        o = torch.matmul(w, v).transpose(1, 2).contiguous().view(b, t, c)
        o = self.to_out[0](o)
        return self.to_out[1](o)


class GEGLU(nn.Module):
    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.proj = nn.Linear(d_in, 2 * d_out)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        return self.net[2](self.net[1](self.net[0](x)))


class TransformerLayer(nn.Module):
    """LN-self-attn, LN-cross-attn, LN-GEGLU-FF, each residual (unet_pt.py:179-210)."""

    def __init__(self, dim: int, head_dim: int, cross_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, head_dim, cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    """GN -> tokens -> proj_in -> layers -> proj_out -> image + skip (unet_pt.py:213-243)."""

    def __init__(self, dim: int, depth: int, head_dim: int, cross_dim: int, groups: int):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList(
            [TransformerLayer(dim, head_dim, cross_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim)

    def forward(self, x, context):
        b, c, h, w = x.shape
        y = self.norm(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
        y = self.proj_in(y)
        for blk in self.transformer_blocks:
            y = blk(y, context)
        y = self.proj_out(y)
        y = y.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
        return y + x


class Downsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Stage(nn.Module):
    """One resolution level of the encoder or decoder.  Attribute names
    (`resnets`, `attentions`, `downsamplers`, `upsamplers`) give the Diffusers
    state_dict keys."""

    def __init__(self, res_io: Sequence[Tuple[int, int]], spec: UNetSpec, depth: int,
                 down: bool = False, up: bool = False):
        super().__init__()
        c_out = res_io[0][1]
        self.resnets = nn.ModuleList([ResBlock(i, o, spec.temb_dim, spec.groups) for i, o in res_io])
        if depth > 0:
            self.attentions = nn.ModuleList(
                [SpatialTransformer(c_out, depth, spec.head_dim, spec.cross_dim, spec.groups)
                 for _ in res_io])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample(c_out)]) if down else None
        self.upsamplers = nn.ModuleList([Upsample(c_out)]) if up else None

    def encode(self, x, temb, context, skips: List):
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, context)
            skips.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            skips.append(x)
        return x

    def decode(self, x, temb, context, skips: List):
        for i, res in enumerate(self.resnets):
            x = res(torch.cat([x, skips.pop()], dim=1), temb)
            if self.attentions is not None:
                x = self.attentions[i](x, context)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class MidStage(nn.Module):
    def __init__(self, c: int, spec: UNetSpec, depth: int):
        super().__init__()
        self.attentions = nn.ModuleList(
            [SpatialTransformer(c, depth, spec.head_dim, spec.cross_dim, spec.groups)])
        self.resnets = nn.ModuleList([ResBlock(c, c, spec.temb_dim, spec.groups) for _ in range(2)])

    def forward(self, x, temb, context):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, context)
        return self.resnets[1](x, temb)


class UNet2DConditionModel(nn.Module):
    """forward(sample, timesteps, encoder_hidden_states, added_cond_kwargs, **ignored) -> [sample]
    (same call signature and list return as reference unet_pt.py:469-542)."""

    def __init__(self, spec: UNetSpec = SDXL_BASE):
        super().__init__()
        self.spec = spec
        self.config = make_config(spec)
        w = spec.widths
        n = len(w)
        self.conv_in = nn.Conv2d(spec.in_channels, w[0], 3, padding=1)
        self.time_proj = SinusoidalProj(spec.time_proj_dim)
        self.time_embedding = TimestepMLP(spec.time_proj_dim, spec.temb_dim)
        self.add_time_proj = SinusoidalProj(spec.add_time_proj_dim)
        self.add_embedding = TimestepMLP(spec.add_in_dim, spec.temb_dim)

        # encoder; record the channel count of every skip it emits
        skip_ch = [w[0]]
        downs = []
        c_prev = w[0]
        for lvl in range(n):
            io = [(c_prev if j == 0 else w[lvl], w[lvl]) for j in range(spec.resnets_per_level)]
            downs.append(Stage(io, spec, spec.depths[lvl], down=lvl < n - 1))
            skip_ch += [w[lvl]] * spec.resnets_per_level + ([w[lvl]] if lvl < n - 1 else [])
            c_prev = w[lvl]
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidStage(w[-1], spec, spec.depths[-1] if spec.mid_depth is None else spec.mid_depth)

        ups = []
        for lvl in reversed(range(n)):
            io = []
            for _ in range(spec.resnets_per_level + 1):
                io.append((c_prev + skip_ch.pop(), w[lvl]))
                c_prev = w[lvl]
            ups.append(Stage(io, spec, spec.depths[lvl], up=lvl > 0))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(spec.groups, w[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(w[0], spec.out_channels, 3, padding=1)

    def forward(self, sample, timesteps, encoder_hidden_states, added_cond_kwargs, **kwargs):
        text_embeds = added_cond_kwargs.get("text_embeds")
        time_ids = added_cond_kwargs.get("time_ids")
        tid = self.add_time_proj(time_ids.flatten()).reshape((text_embeds.shape[0], -1))
        return self.denoise(sample, timesteps, encoder_hidden_states, torch.concat([text_embeds, tid], dim=-1))

    def denoise(self, sample, timesteps, encoder_hidden_states, add):
        """`add` = pooled text embedding | Fourier features of the size/crop ids, (B, add_in_dim)."""
        timesteps = timesteps.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(timesteps).to(dtype=sample.dtype))
        emb = emb + self.add_embedding(add.to(emb.dtype))

        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x = blk.encode(x, emb, encoder_hidden_states, skips)
        x = self.mid_block(x, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk.decode(x, emb, encoder_hidden_states, skips)
        x = self.conv_out(self.conv_act(self.conv_norm_out(x)))
        return [x]


class UNetWithLabelVector(nn.Module):
    """The same network entered the way ComfyUI calls an SDXL UNet: `y` is the ready-made (B, add_in_dim) vector
    (pooled text embedding | Fourier features of the six size/crop ids, each through the 256-wide cos|sin embedder -
    exactly the `add` vector the Diffusers form builds from added_cond_kwargs).  Shares its weights with `unet`."""

    def __init__(self, unet: "UNet2DConditionModel"):
        super().__init__()
        self.unet = unet

    def forward(self, sample, timesteps, encoder_hidden_states, y, **kwargs):
        return self.unet.denoise(sample, timesteps, encoder_hidden_states, y)


def make_config(spec: UNetSpec):
    """The three attributes the Diffusers SDXL pipeline reads from `unet.config`
    (reference unet_pt.py:423-428; re-attached after tracing in
    implementations/Diffusers/load_sdxl_pipeline.py:29-34)."""
    cfg = namedtuple("config", "in_channels addition_time_embed_dim sample_size")
    return cfg(spec.in_channels, spec.add_time_proj_dim, spec.sample_size)
