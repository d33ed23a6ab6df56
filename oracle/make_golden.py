"""TEST INFRASTRUCTURE - generates tests/golden/*.npz from the REFERENCE itself.

Runs only in the build container (needs /root/reference).  It imports the
reference's eager model file standalone (it depends on torch only; the package
__init__ pulls xformers/diffusers, which are absent - SURVEY.md 8c), loads the
build's synthetic weights into it by parameter name and records outputs.  The
fixtures are data (inputs are regenerated from seeds; outputs are stored); no
reference source is copied.

    python oracle/make_golden.py f1 f2          # seconds .. a minute
    python oracle/make_golden.py f3_64          # ~4 min   (50 Euler steps, latent 64)
    python oracle/make_golden.py f3_128         # ~20 min  (50 Euler steps, latent 128)
    python oracle/make_golden.py f1_b4          # ~1 min   (BASELINE config #3 rows: batch 4, per-row distinct conditioning)
    python oracle/make_golden.py f3_b2          # ~8 min   (two independent 50-step trajectories in one batch, latent 64)
    python oracle/make_golden.py f3_cfg         # ~8 min   (the Diffusers call-site protocol: CFG batch 2, 50 steps, latent 64)
    python oracle/make_golden.py f1_b4_128      # ~3 min   (BASELINE config #3 at its own size: batch 4, latent 128, one step; stored sub-sampled)
    python oracle/make_golden.py f2_large       # ~1 min   (SURVEY 8c's attention sizes: T=1024 @ C=1280, T=4096 @ C=640; GroupNorm at 128 x 128)
    python oracle/make_golden.py f3_64_f64 f3_cfg_f64   # the reference module in DOUBLE precision through the same loops: what
                                                # the reference's own fp32 arithmetic deviates from (the noise floor the 1e-3 gates are read against)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stabletriton_amd import synth                     # noqa: E402
from stabletriton_amd.scheduler import euler_discrete_tables  # noqa: E402
from oracle import unet_oracle as orc                  # noqa: E402

REF_FILE = "/root/reference/src/stabletriton/optimizers/unet_pt.py"
OUT = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED, INPUT_SEED = 0, 1234


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_unet_pt", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def meta():
    return dict(torch_version=torch.__version__, weight_seed=WEIGHT_SEED, input_seed=INPUT_SEED,
                generated=time.strftime("%Y-%m-%d"))


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()}, **{"meta_" + k: np.asarray(v) for k, v in meta().items()})
    print("wrote", path, os.path.getsize(path), "bytes")


F2_STRIDE = 31          # prime, so the kept elements walk across rows and columns


def subsample(t: torch.Tensor) -> torch.Tensor:
    """Large per-op outputs are stored as every 31st element of the flattened
    tensor (tests apply the same rule to their own output)."""
    return t.flatten()[::F2_STRIDE].clone() if t.numel() > 20000 else t


_REF_UNET = None


def ref_unet(ref):
    global _REF_UNET
    if _REF_UNET is None:
        t0 = time.time()
        m = ref.UNet2DConditionModel().eval()
        synth.fill_module_(m, WEIGHT_SEED)
        print(f"reference UNet built + filled in {time.time() - t0:.0f}s; params",
              sum(p.numel() for p in m.parameters()))
        _REF_UNET = m
    return _REF_UNET


@torch.no_grad()
def f1(ref):
    """BASELINE config #1: one eager CPU fp32 step at 512x512 (latent 64)."""
    m = ref_unet(ref)
    x = synth.denoise_inputs(1, 64, INPUT_SEED)
    t = torch.tensor(999.0)
    out = m(x["latent"], t, x["encoder_hidden_states"],
            {"text_embeds": x["text_embeds"], "time_ids": x["time_ids"]})[0]
    # cross-check the restatement right here, on the same weights
    sd = {k: v for k, v in m.state_dict().items()}
    mine = orc.unet_forward(sd, x["latent"], t, x["encoder_hidden_states"], x["text_embeds"], x["time_ids"])
    print("F1 |oracle - reference| max =", float((mine - out).abs().max()), " |ref| max =", float(out.abs().max()))
    save("f1_unet_step_latent64", out=out, timestep=999.0, latent_hw=64)


@torch.no_grad()
def f2(ref):
    """Per-op fixtures at real SDXL widths (outputs of the reference's own sub-modules)."""
    arrays = {}

    def filled(mod, prefix):
        for n, p in mod.named_parameters():
            p.copy_(synth.param_tensor(f"{prefix}.{n}", tuple(p.shape), WEIGHT_SEED))
        return mod.eval()

    # attention, self (T=256) and cross (S=77), C = 640 and 1280 (unet_pt.py:98-147)
    for c in (640, 1280):
        a = filled(ref.Attention(c), f"f2.attn_self{c}")
        x = synth.normal(f"f2.attn_self{c}.x", (1, 256, c), INPUT_SEED)
        arrays[f"attn_self{c}"] = a(x)
        a = filled(ref.Attention(c, 2048), f"f2.attn_cross{c}")
        ctx = synth.normal(f"f2.attn_cross{c}.ctx", (1, 77, 2048), INPUT_SEED)
        arrays[f"attn_cross{c}"] = a(x, ctx)
    # resnet blocks (unet_pt.py:54-95)
    for cin, cout, sc in ((320, 320, False), (960, 320, True)):
        r = filled(ref.ResnetBlock2D(cin, cout, conv_shortcut=sc), f"f2.res{cin}_{cout}")
        x = synth.normal(f"f2.res{cin}_{cout}.x", (1, cin, 16, 16), INPUT_SEED)
        temb = synth.normal(f"f2.res{cin}_{cout}.temb", (1, 1280), INPUT_SEED)
        arrays[f"res{cin}_{cout}"] = r(x, temb)
    # GEGLU (unet_pt.py:150-158)
    g = filled(ref.GEGLU(640, 2560), "f2.geglu")
    arrays["geglu"] = g(synth.normal("f2.geglu.x", (1, 64, 640), INPUT_SEED))
    # spatial transformer, depth 1 (unet_pt.py:213-243)
    tr = filled(ref.Transformer2DModel(640, 640, 1), "f2.xfmr")
    x = synth.normal("f2.xfmr.x", (1, 640, 16, 16), INPUT_SEED)
    ctx = synth.normal("f2.xfmr.ctx", (1, 77, 2048), INPUT_SEED)
    arrays["xfmr"] = tr(x, ctx)
    # sinusoidal features (unet_pt.py:17-36)
    tt = torch.tensor([999.0, 500.0, 1.0, 1024.0, 0.0])
    arrays["timesteps320"] = ref.Timesteps(320)(tt)
    arrays["timesteps256"] = ref.Timesteps(256)(tt)
    # GroupNorm at the non-power-of-two group sizes SDXL uses (C/32 = 10,20,30,40,60,80)
    for c, eps in ((320, 1e-5), (640, 1e-6), (960, 1e-5), (1280, 1e-6), (1920, 1e-5), (2560, 1e-5)):
        gn = filled(torch.nn.GroupNorm(32, c, eps=eps), f"f2.gn{c}")
        x = synth.normal(f"f2.gn{c}.x", (1, c, 8, 8), INPUT_SEED)
        arrays[f"gn{c}"] = gn(x)
    save("f2_ops", **{k: subsample(v) for k, v in arrays.items()})


@torch.no_grad()
def f3(ref, hw):
    """50-step Euler loop driving the reference UNet (same loop code as the build uses)."""
    m = ref_unet(ref)
    x = synth.denoise_inputs(1, hw, INPUT_SEED)
    tables = euler_discrete_tables(50)
    cond = {"text_embeds": x["text_embeds"], "time_ids": x["time_ids"]}
    t0 = time.time()
    trace = []

    def fn(x_in, t):
        e = m(x_in, t, x["encoder_hidden_states"], cond)[0]
        trace.append(float(e.abs().mean()))
        if len(trace) % 5 == 0:
            print(f"  step {len(trace)} |eps| mean {trace[-1]:.4f}  {time.time() - t0:.0f}s", flush=True)
        return e

    final = orc.euler_denoise(fn, x["latent"], tables)
    save(f"f3_euler50_latent{hw}", final=final, eps_abs_mean=np.asarray(trace, dtype=np.float32), latent_hw=hw)


@torch.no_grad()
def f1_b4(ref):
    """BASELINE config #3 (bs=4, 77-token text conditioning, per-row distinct): one eager step of the
    reference at latent 64 with a scalar timestep (reference: `timesteps.expand(batch)`, unet_pt.py:473)
    and one with a different timestep per row (ComfyUI-style callers pass a (B,) tensor)."""
    m = ref_unet(ref)
    x = synth.denoise_inputs(4, 64, INPUT_SEED)
    cond = {"text_embeds": x["text_embeds"], "time_ids": x["time_ids"]}
    out = m(x["latent"], torch.tensor(500.0), x["encoder_hidden_states"], cond)[0]
    tvec = torch.tensor([999.0, 700.0, 400.0, 100.0])
    out_tvec = m(x["latent"], tvec, x["encoder_hidden_states"], cond)[0]
    sd = {k: v for k, v in m.state_dict().items()}
    mine = orc.unet_forward(sd, x["latent"], torch.tensor(500.0), x["encoder_hidden_states"], x["text_embeds"], x["time_ids"])
    print("F1_b4 |oracle - reference| max =", float((mine - out).abs().max()), " |ref| max =", float(out.abs().max()))
    mine = orc.unet_forward(sd, x["latent"], tvec, x["encoder_hidden_states"], x["text_embeds"], x["time_ids"])
    print("F1_b4(tvec) |oracle - reference| max =", float((mine - out_tvec).abs().max()))
    save("f1_unet_step_latent64_b4", out=out, out_tvec=out_tvec, timestep=500.0, timesteps_vec=tvec, latent_hw=64, batch=4)


@torch.no_grad()
def f1_b4_128(ref):
    """BASELINE config #3 at the size it is benchmarked at (bs=4, 1024 x 1024 = latent 128, per-row distinct 77-token
    conditioning): one eager step of the reference; every 31st value of the (4, 4, 128, 128) output is kept."""
    m = ref_unet(ref)
    x = synth.denoise_inputs(4, 128, INPUT_SEED)
    cond = {"text_embeds": x["text_embeds"], "time_ids": x["time_ids"]}
    t0 = time.time()
    out = m(x["latent"], torch.tensor(500.0), x["encoder_hidden_states"], cond)[0]
    print(f"F1_b4_128: {time.time() - t0:.0f}s  |ref| max = {float(out.abs().max()):.3f} rms = {float(out.pow(2).mean().sqrt()):.3f}")
    save("f1_unet_step_latent128_b4", out=subsample(out), timestep=500.0, latent_hw=128, batch=4, out_rms=float(out.pow(2).mean().sqrt()),
         out_max_abs=float(out.abs().max()))


@torch.no_grad()
def f1_rect(ref, h=96, w=64):
    """A RECTANGULAR latent (768 x 512 px = 96 x 64: the reference's convolutions and its token-count-agnostic attention take any
    size whose sides the two Downsample2D / Upsample2D pairs bring back, unet_pt.py:246-266): one eager step of the reference,
    every 31st value of the (1, 4, 96, 64) output; the restatement is cross-checked on the full output."""
    m = ref_unet(ref)
    x = synth.denoise_inputs(1, (h, w), INPUT_SEED)
    cond = {"text_embeds": x["text_embeds"], "time_ids": x["time_ids"]}
    t = torch.tensor(500.0)
    t0 = time.time()
    out = m(x["latent"], t, x["encoder_hidden_states"], cond)[0]
    sd = {k: v for k, v in m.state_dict().items()}
    mine = orc.unet_forward(sd, x["latent"], t, x["encoder_hidden_states"], x["text_embeds"], x["time_ids"])
    print(f"F1_rect {h}x{w}: {time.time() - t0:.0f}s  |oracle - reference| max = {float((mine - out).abs().max()):.2e}  |ref| max = {float(out.abs().max()):.3f}")
    save(f"f1_unet_step_latent{h}x{w}", out=subsample(out), timestep=500.0, latent_h=h, latent_w=w, out_rms=float(out.pow(2).mean().sqrt()),
         out_max_abs=float(out.abs().max()))


@torch.no_grad()
def f3_b2(ref, hw=64):
    """Two independent prompts in one batch through the 50-step Euler loop (the batched DenoiseLoop)."""
    m = ref_unet(ref)
    x = synth.denoise_inputs(2, hw, INPUT_SEED)
    tables = euler_discrete_tables(50)
    cond = {"text_embeds": x["text_embeds"], "time_ids": x["time_ids"]}
    t0 = time.time()
    n = [0]

    def fn(x_in, t):
        n[0] += 1
        if n[0] % 5 == 0:
            print(f"  step {n[0]}  {time.time() - t0:.0f}s", flush=True)
        return m(x_in, t, x["encoder_hidden_states"], cond)[0]

    final = orc.euler_denoise(fn, x["latent"], tables)
    save(f"f3_euler50_latent{hw}_b2", final=final, latent_hw=hw, batch=2)


CFG_SCALE = 5.0           # SDXL pipeline default guidance_scale


@torch.no_grad()
def f3_cfg(ref, hw=64):
    """The protocol of the reference's call site (implementations/Diffusers/load_sdxl_pipeline.py:39-46 ->
    diffusers 0.21.2 SDXL pipeline): ONE latent, the UNet sees batch 2 = [negative, positive] conditioning
    every step, eps = eps_neg + g * (eps_pos - eps_neg), Euler update.  The loop code is orc.euler_denoise_cfg
    on both sides."""
    m = ref_unet(ref)
    x = synth.denoise_inputs(2, hw, INPUT_SEED)          # row 0 = negative prompt, row 1 = prompt
    tables = euler_discrete_tables(50)
    cond = {"text_embeds": x["text_embeds"], "time_ids": x["time_ids"]}
    t0 = time.time()
    n = [0]

    def fn(x_in2, t):
        n[0] += 1
        if n[0] % 5 == 0:
            print(f"  step {n[0]}  {time.time() - t0:.0f}s", flush=True)
        return m(x_in2, t, x["encoder_hidden_states"], cond)[0]

    final = orc.euler_denoise_cfg(fn, x["latent"][:1], tables, CFG_SCALE)
    save(f"f3_cfg50_latent{hw}", final=final, latent_hw=hw, guidance_scale=CFG_SCALE)


@torch.no_grad()
def f2_large(ref):
    """SURVEY 8(c)'s F2 attention sizes - self (T=1024, 20 heads) and text-context (S=77) at C=1280, self (T=4096, 10 heads) and
    text-context at C=640 (unet_pt.py:98-147) - and GroupNorm on the 128 x 128 feature maps whose groups do not fit one block
    (960 channels: 491,520 elements per group).  Stored sub-sampled like f2_ops, in a file of their own."""
    arrays = {}

    def filled(mod, prefix):
        for n, p in mod.named_parameters():
            p.copy_(synth.param_tensor(f"{prefix}.{n}", tuple(p.shape), WEIGHT_SEED))
        return mod.eval()

    for c, t in ((1280, 1024), (640, 4096)):
        a = filled(ref.Attention(c), f"f2.attn_self{c}_T{t}")
        x = synth.normal(f"f2.attn_self{c}_T{t}.x", (1, t, c), INPUT_SEED)
        arrays[f"attn_self{c}_T{t}"] = a(x)
        a = filled(ref.Attention(c, 2048), f"f2.attn_cross{c}_T{t}")
        ctx = synth.normal(f"f2.attn_cross{c}_T{t}.ctx", (1, 77, 2048), INPUT_SEED)
        arrays[f"attn_cross{c}_T{t}"] = a(x, ctx)
    for c, eps in ((960, 1e-5), (320, 1e-5), (640, 1e-6)):
        gn = filled(torch.nn.GroupNorm(32, c, eps=eps), f"f2.gn{c}_128")
        x = synth.normal(f"f2.gn{c}_128.x", (1, c, 128, 128), INPUT_SEED)
        arrays[f"gn{c}_128"] = gn(x)
    save("f2_ops_large", **{k: subsample(v) for k, v in arrays.items()})


@torch.no_grad()
def f3_f64(ref, cfg: bool, hw=64):
    """The reference module converted to DOUBLE (`.double()`: the same fp32 weight values, float64 arithmetic) through the same
    loop with the same fp32 table constants.  Recorded beside it: how far the reference's own fp32 run (the committed golden the
    strict gates compare with) is from this - the noise floor of any fp32 summation order; a gate tighter than it measures luck."""
    name = f"f3_cfg50_latent{hw}" if cfg else f"f3_euler50_latent{hw}"
    ref32 = torch.from_numpy(np.load(os.path.join(OUT, name + ".npz"))["final"])
    m = ref_unet(ref).double()
    x = synth.denoise_inputs(2 if cfg else 1, hw, INPUT_SEED)
    ehs = x["encoder_hidden_states"].double()
    cond = {"text_embeds": x["text_embeds"].double(), "time_ids": x["time_ids"].double()}
    tables = euler_discrete_tables(50)
    t0 = time.time()
    n = [0]

    def fn(x_in, t):
        n[0] += 1
        if n[0] % 5 == 0:
            print(f"  step {n[0]}  {time.time() - t0:.0f}s", flush=True)
        return m(x_in, t, ehs, cond)[0]

    if cfg:
        final = orc.euler_denoise_cfg(fn, x["latent"][:1], tables, CFG_SCALE, state_dtype=torch.float64)
    else:
        final = orc.euler_denoise(fn, x["latent"], tables, state_dtype=torch.float64)
    dev = (ref32.double() - final).abs()
    print(f"{name}: reference fp32 vs float64: max abs {float(dev.max()):.3e} rms {float(dev.pow(2).mean().sqrt()):.3e}"
          f"  |latent| max {float(final.abs().max()):.2f}")
    save(name + "_f64", final=final, ref_fp32_max_abs=float(dev.max()), ref_fp32_rms=float(dev.pow(2).mean().sqrt()), latent_hw=hw,
         **({"guidance_scale": CFG_SCALE} if cfg else {}))
    global _REF_UNET
    _REF_UNET = None            # the module is double now: later fixtures of this process build a fresh one


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    ref = load_reference()
    for what in sys.argv[1:] or ["f1", "f2"]:
        if what == "f1":
            f1(ref)
        elif what == "f2":
            f2(ref)
        elif what == "f1_b4":
            f1_b4(ref)
        elif what == "f3_b2":
            f3_b2(ref)
        elif what == "f1_b4_128":
            f1_b4_128(ref)
        elif what == "f1_rect":
            f1_rect(ref)
        elif what == "f3_cfg":
            f3_cfg(ref)
        elif what == "f2_large":
            f2_large(ref)
        elif what == "f3_64_f64":
            f3_f64(ref, cfg=False)
        elif what == "f3_cfg_f64":
            f3_f64(ref, cfg=True)
        elif what.startswith("f3_"):
            f3(ref, int(what[3:]))
        else:
            raise SystemExit(f"unknown fixture {what}")
