"""End-to-end parity of the compiled UNet and the captured denoise loop.

fp32 ("strict") runs must meet the north_star bound: <= 1e-3 abs on the output /
final latent against the reference's eager path (golden fixtures generated from
the reference itself, oracle/make_golden.py).  bf16 runs use the same kernels
with 8-bit mantissas; their deviation is reported and bounded loosely.  fp16 runs
(the reference's own compute type: its call site passes a .half() module) are
bounded at a quarter of the bf16 bounds."""
import numpy as np
import pytest
import torch

from oracle import unet_oracle as orc
from stabletriton_amd import synth
from stabletriton_amd.optimization import optimize_model
from stabletriton_amd.pipeline import DenoiseLoop
from stabletriton_amd.scheduler import euler_discrete_tables
from stabletriton_amd.unet import SDXL_BASE, TINY, UNet2DConditionModel
from stabletriton_amd import unet as U
from tests.util import golden, rel_err

pytestmark = pytest.mark.gpu
ABS_TOL_STRICT = 1e-3          # north_star: 1e-3 abs on the final latent
# bf16 / fp16 runs are bounded by what 16-bit STORAGE costs by itself: oracle/make_rounded_golden.py ran the (reference-pinned)
# oracle on the same inputs with every weight and every operator result rounded to the type and fp32 arithmetic inside the
# operators, and recorded its deviation from the reference's fp32 output (tests/golden/*_rounded.npz: F1 bf16 max abs 6.2e-2 /
# rms 1.8e-2, fp16 8.3e-3 / 2.2e-3; F3 latent 64 bf16 0.36 / 8.8e-2).  The HIP path's own deviation may be at most
# STORAGE_FACTOR times that: two realisations of the same rounding noise differ, a kernel that adds error of its own shows.
STORAGE_FACTOR = 1.3          # (round 5, ADVICE r4: 2.0 let a kernel regression that doubles the error pass; measured / storage-alone is 0.95-1.05)


def storage_bound(name, dt):
    """(max abs, rms) deviation from the reference that `dt` storage alone causes on fixture `name`, times STORAGE_FACTOR"""
    g = golden(name + "_rounded")
    return STORAGE_FACTOR * float(g[dt + "_max_abs"]), STORAGE_FACTOR * float(g[dt + "_rms"])
F2_STRIDE = 31                 # oracle/make_golden.py subsample rule


def build(spec, dtype, dev, graph):
    with torch.device("meta"):
        m = UNet2DConditionModel(spec)
    m = m.to_empty(device=dev).to(dtype).eval().requires_grad_(False)
    synth.fill_module_(m, 0)
    return m, optimize_model(m, cuda_graph=graph)


def tiny_inputs(dtype, dev, batch=1, hw=16):
    x = synth.denoise_inputs(batch, hw, 1234, cross_dim=TINY.cross_dim, pooled_dim=TINY.pooled_dim)
    return x, {k: v.to(dev, dtype) for k, v in x.items()}


@pytest.mark.parametrize("dtype,tol", [(torch.float32, ABS_TOL_STRICT), (torch.bfloat16, 0.1), (torch.float16, 0.025)])
@pytest.mark.parametrize("batch,hw", [(1, 16), (2, 8), (1, 24), (1, (16, 24)), (2, (24, 8)), (1, (12, 20)), (1, (8, 40))])      # (h, w): rectangular latents
def test_tiny_unet_step(gpu, dtype, tol, batch, hw):
    m, gm = build(TINY, dtype, gpu, graph=False)
    sd = {k: v.float().cpu() for k, v in m.state_dict().items()}
    x, xg = tiny_inputs(dtype, gpu, batch, hw)
    xr = {k: v.to(dtype).float() for k, v in x.items()}
    t = torch.tensor(321.0)
    ref = orc.unet_forward(sd, xr["latent"], t, xr["encoder_hidden_states"], xr["text_embeds"], xr["time_ids"])
    with torch.no_grad():
        out = gm(xg["latent"], t.to(gpu), xg["encoder_hidden_states"],
                 {"text_embeds": xg["text_embeds"], "time_ids": xg["time_ids"]})[0]
    err = float((out.float().cpu() - ref).abs().max())
    print(f"tiny step {dtype} b{batch} hw{hw}: max abs err {err:.2e}, |ref|max {float(ref.abs().max()):.2f}")
    assert err <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, ABS_TOL_STRICT), (torch.bfloat16, 0.1), (torch.float16, 0.025)])
@pytest.mark.parametrize("head_dim", [32, 128])
def test_tiny_unet_step_other_head_sizes(gpu, dtype, tol, head_dim):
    """A model with the reference's other head sizes (kernels/attention_fa2.py:118-123) compiles through the same passes:
    fuse_attention matches, the text-context query path takes the two-launch route (the fused one is head_dim 64 only),
    the generic attention kernel runs; strict mode inside the same 1e-3."""
    import dataclasses
    spec = dataclasses.replace(TINY, head_dim=head_dim)
    m, gm = build(spec, dtype, gpu, graph=False)
    sd = {k: v.float().cpu() for k, v in m.state_dict().items()}
    x, xg = tiny_inputs(dtype, gpu, 2, 16)
    xr = {k: v.to(dtype).float() for k, v in x.items()}
    t = torch.tensor(500.0)
    ref = orc.unet_forward(sd, xr["latent"], t, xr["encoder_hidden_states"], xr["text_embeds"], xr["time_ids"], head_dim=head_dim)
    with torch.no_grad():
        out = gm(xg["latent"], t.to(gpu), xg["encoder_hidden_states"], {"text_embeds": xg["text_embeds"], "time_ids": xg["time_ids"]})[0]
    err = float((out.float().cpu() - ref).abs().max())
    print(f"tiny step head_dim {head_dim} {dtype}: max abs err {err:.2e}")
    assert err <= tol


def test_graph_cache_replays_and_rekeys(gpu):
    m, gm_plain = build(TINY, torch.float32, gpu, graph=False)
    gm = optimize_model(m, cuda_graph=True)
    t = torch.tensor(10.0, device=gpu)
    outs = []
    for batch in (1, 1, 2, 1):
        _, xg = tiny_inputs(torch.float32, gpu, batch, 16)
        xg["latent"] = xg["latent"] * (1 + len(outs))          # new values, same shapes
        cond = {"text_embeds": xg["text_embeds"], "time_ids": xg["time_ids"]}
        with torch.no_grad():
            a = gm(xg["latent"], t, xg["encoder_hidden_states"], cond)[0]
            b = gm_plain(xg["latent"], t, xg["encoder_hidden_states"], cond)[0]
        assert torch.equal(a, b), "graph replay must reproduce the eager launch sequence bit for bit"
        outs.append(a)
    assert len(gm.forward._cached) == 2                         # batch 1 and batch 2, never keyed on values
    # a CPU timestep value does not create new graphs (reference defect: graphs.py:197-199)
    _, xg = tiny_inputs(torch.float32, gpu, 1, 16)
    cond = {"text_embeds": xg["text_embeds"], "time_ids": xg["time_ids"]}
    n0 = len(gm.forward._cached)
    with torch.no_grad():
        for tv in (1.0, 2.0, 3.0):
            gm(xg["latent"], torch.tensor(tv), xg["encoder_hidden_states"], cond)
    assert len(gm.forward._cached) == n0 + 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_tiny_denoise_loop_modes(gpu, dtype):
    m, gm = build(TINY, dtype, gpu, graph=False)
    tables = euler_discrete_tables(10)
    x, xg = tiny_inputs(dtype, gpu, 1, 16)
    finals = {}
    for mode in ("eager", "step", "loop"):
        loop = DenoiseLoop(gm, 1, 16, dtype, gpu, tables, cross_dim=TINY.cross_dim, pooled_dim=TINY.pooled_dim, mode=mode)
        loop.set_conditioning(xg["encoder_hidden_states"], xg["text_embeds"], xg["time_ids"])
        with torch.no_grad():
            finals[mode] = loop.denoise(x["latent"])
            again = loop.denoise(x["latent"])                   # replay of the same graph
        assert torch.equal(finals[mode], again)
    assert torch.equal(finals["eager"], finals["step"]) and torch.equal(finals["eager"], finals["loop"])
    if dtype == torch.float32:
        sd = {k: v.float().cpu() for k, v in m.state_dict().items()}
        ref = orc.euler_denoise(
            lambda xi, t: orc.unet_forward(sd, xi, t, x["encoder_hidden_states"], x["text_embeds"], x["time_ids"]),
            x["latent"], tables)
        err = float((finals["loop"].cpu() - ref).abs().max())
        print(f"tiny 10-step loop fp32: max abs err on final latent {err:.2e}")
        assert err <= ABS_TOL_STRICT


# ---------------------------------------------------------------------------------- F2: per-op, reference-generated
def _filled(mod, prefix, dtype, dev):
    mod = mod.eval().requires_grad_(False)
    for n, p in mod.named_parameters():
        p.copy_(synth.param_tensor(f"{prefix}.{n}", tuple(p.shape), 0))
    return mod.to(dev, dtype)


def _sub(t):
    return t.flatten()[::F2_STRIDE] if t.numel() > 20000 else t


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2), (torch.float16, 7.5e-3)])
def test_f2_golden_ops(gpu, dtype, tol):
    g = golden("f2_ops")
    seed = 1234

    def check(name, out):
        ref = torch.from_numpy(g[name])
        e = rel_err(_sub(out.float().cpu()).reshape(ref.shape), ref)
        print(f"F2 {name} {dtype}: rel err {e:.2e}")
        assert e <= tol, name

    class SelfAttn(torch.nn.Module):
        def __init__(self, c):
            super().__init__()
            self.attn = U.Attention(c, 64)

        def forward(self, x):
            return self.attn(x)

    class CrossAttn(torch.nn.Module):
        def __init__(self, c):
            super().__init__()
            self.attn = U.Attention(c, 64, 2048)

        def forward(self, x, ctx):
            return self.attn(x, ctx)

    for c in (640, 1280):
        x = synth.normal(f"f2.attn_self{c}.x", (1, 256, c), seed).to(gpu, dtype)
        h = SelfAttn(c)
        _filled(h.attn, f"f2.attn_self{c}", dtype, gpu)
        check(f"attn_self{c}", optimize_model(h.to(gpu, dtype), False)(x))
        h = CrossAttn(c)
        _filled(h.attn, f"f2.attn_cross{c}", dtype, gpu)
        ctx = synth.normal(f"f2.attn_cross{c}.ctx", (1, 77, 2048), seed).to(gpu, dtype)
        check(f"attn_cross{c}", optimize_model(h.to(gpu, dtype), False)(x, ctx))
    for cin, cout in ((320, 320), (960, 320)):
        r = _filled(U.ResBlock(cin, cout, 1280, 32), f"f2.res{cin}_{cout}", dtype, gpu)
        x = synth.normal(f"f2.res{cin}_{cout}.x", (1, cin, 16, 16), seed).to(gpu, dtype)
        temb = synth.normal(f"f2.res{cin}_{cout}.temb", (1, 1280), seed).to(gpu, dtype)
        check(f"res{cin}_{cout}", optimize_model(r, False)(x, temb))
    ge = _filled(U.GEGLU(640, 2560), "f2.geglu", dtype, gpu)
    check("geglu", optimize_model(ge, False)(synth.normal("f2.geglu.x", (1, 64, 640), seed).to(gpu, dtype)))
    tr = _filled(U.SpatialTransformer(640, 1, 64, 2048, 32), "f2.xfmr", dtype, gpu)
    x = synth.normal("f2.xfmr.x", (1, 640, 16, 16), seed).to(gpu, dtype)
    ctx = synth.normal("f2.xfmr.ctx", (1, 77, 2048), seed).to(gpu, dtype)
    check("xfmr", optimize_model(tr, False)(x, ctx))
    for c, eps in ((320, 1e-5), (640, 1e-6), (960, 1e-5), (1280, 1e-6), (1920, 1e-5), (2560, 1e-5)):
        gn = _filled(torch.nn.GroupNorm(32, c, eps=eps), f"f2.gn{c}", dtype, gpu)
        x = synth.normal(f"f2.gn{c}.x", (1, c, 8, 8), seed).to(gpu, dtype)
        wrap = torch.nn.Sequential(gn)
        check(f"gn{c}", optimize_model(wrap, False)(x))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2), (torch.float16, 7.5e-3)])
def test_f2_large_golden_ops(gpu, dtype, tol):
    """The reference's own outputs at the sizes SURVEY 8(c) names (round 5 fixtures): self-attention at 1024 tokens x 20 heads and
    4096 tokens x 10 heads (the site that was only reached through the 50-step latent before), the text-context attention at
    both widths, GroupNorm on the 128 x 128 maps (groups of up to 491,520 elements)."""
    g = golden("f2_ops_large")
    seed = 1234

    def check(name, out):
        ref = torch.from_numpy(g[name])
        e = rel_err(_sub(out.float().cpu()).reshape(ref.shape), ref)
        print(f"F2-large {name} {dtype}: rel err {e:.2e}")
        assert e <= tol, name

    class SelfAttn(torch.nn.Module):
        def __init__(self, c):
            super().__init__()
            self.attn = U.Attention(c, 64)

        def forward(self, x):
            return self.attn(x)

    class CrossAttn(torch.nn.Module):
        def __init__(self, c):
            super().__init__()
            self.attn = U.Attention(c, 64, 2048)

        def forward(self, x, ctx):
            return self.attn(x, ctx)

    for c, t in ((1280, 1024), (640, 4096)):
        x = synth.normal(f"f2.attn_self{c}_T{t}.x", (1, t, c), seed).to(gpu, dtype)
        h = SelfAttn(c)
        _filled(h.attn, f"f2.attn_self{c}_T{t}", dtype, gpu)
        check(f"attn_self{c}_T{t}", optimize_model(h.to(gpu, dtype), False)(x))
        h = CrossAttn(c)
        _filled(h.attn, f"f2.attn_cross{c}_T{t}", dtype, gpu)
        ctx = synth.normal(f"f2.attn_cross{c}_T{t}.ctx", (1, 77, 2048), seed).to(gpu, dtype)
        check(f"attn_cross{c}_T{t}", optimize_model(h.to(gpu, dtype), False)(x, ctx))
    for c, eps in ((960, 1e-5), (320, 1e-5), (640, 1e-6)):
        gn = _filled(torch.nn.GroupNorm(32, c, eps=eps), f"f2.gn{c}_128", dtype, gpu)
        x = synth.normal(f"f2.gn{c}_128.x", (1, c, 128, 128), seed).to(gpu, dtype)
        check(f"gn{c}_128", optimize_model(torch.nn.Sequential(gn), False)(x))


# ---------------------------------------------------------------------------------- SDXL-base, reference goldens
# (the SDXL-base fixtures `sdxl_fp32` / `sdxl_bf16` live in conftest.py: one build per session, shared with test_hooks_gpu.py)

def _sdxl_step(gm, dtype, dev, hw):
    x = synth.denoise_inputs(1, hw, 1234)
    xg = {k: v.to(dev, dtype) for k, v in x.items()}
    with torch.no_grad():
        return gm(xg["latent"], torch.tensor(999.0, device=dev), xg["encoder_hidden_states"],
                  {"text_embeds": xg["text_embeds"], "time_ids": xg["time_ids"]})[0].float().cpu()


def test_sdxl_f1_step_fp32(gpu, sdxl_fp32):
    """BASELINE config #1 input through the HIP path vs the reference's own output."""
    ref = torch.from_numpy(golden("f1_unet_step_latent64")["out"])
    out = _sdxl_step(sdxl_fp32, torch.float32, gpu, 64)
    err = float((out - ref).abs().max())
    print(f"F1 fp32: max abs err {err:.2e} (|ref| max {float(ref.abs().max()):.2f})")
    assert err <= ABS_TOL_STRICT


def test_sdxl_f1_step_bf16(gpu, sdxl_bf16):
    ref = torch.from_numpy(golden("f1_unet_step_latent64")["out"])
    out = _sdxl_step(sdxl_bf16, torch.bfloat16, gpu, 64)
    err = float((out - ref).abs().max())
    print(f"F1 bf16: max abs err {err:.2e}, rms err {float((out - ref).pow(2).mean().sqrt()):.2e} "
          f"(|ref| max {float(ref.abs().max()):.2f}, rms {float(ref.pow(2).mean().sqrt()):.2f})")
    mx, rms_b = storage_bound("f1_unet_step_latent64", "bf16")
    assert err <= mx and float((out - ref).pow(2).mean().sqrt()) <= rms_b


def test_sdxl_f1_step_fp16(gpu, sdxl_fp16):
    """The reference call site's own dtype (`UNet2DConditionModelPT().half().cuda()`, load_sdxl_pipeline.py:17-28) through
    `optimize_model`: f16 MFMA kernels end to end, bounded at a quarter of the bf16 bounds."""
    ref = torch.from_numpy(golden("f1_unet_step_latent64")["out"])
    out = _sdxl_step(sdxl_fp16, torch.float16, gpu, 64)
    err, rms = float((out - ref).abs().max()), float((out - ref).pow(2).mean().sqrt())
    print(f"F1 fp16: max abs err {err:.2e}, rms err {rms:.2e} (|ref| max {float(ref.abs().max()):.2f}, rms {float(ref.pow(2).mean().sqrt()):.2f})")
    assert torch.isfinite(out).all()
    mx, rms_b = storage_bound("f1_unet_step_latent64", "fp16")
    assert err <= mx and rms <= rms_b


def _sdxl_loop(gm, dtype, dev, hw, mode="loop"):
    x = synth.denoise_inputs(1, hw, 1234)
    loop = DenoiseLoop(gm, 1, hw, dtype, dev, euler_discrete_tables(50), mode=mode)
    loop.set_conditioning(x["encoder_hidden_states"].to(dev, dtype), x["text_embeds"].to(dev, dtype),
                          x["time_ids"].to(dev, dtype))
    with torch.no_grad():
        return loop.denoise(x["latent"]).cpu()


@pytest.mark.parametrize("hw", [64, 128])
def test_sdxl_f3_euler50_fp32(gpu, sdxl_fp32, hw):
    """50 Euler steps, final latent vs the reference UNet in the same loop (north_star bound)."""
    ref = torch.from_numpy(golden(f"f3_euler50_latent{hw}")["final"])
    out = _sdxl_loop(sdxl_fp32, torch.float32, gpu, hw, mode="step")
    err = float((out - ref).abs().max())
    print(f"F3 latent{hw} fp32: max abs err on final latent {err:.2e} (|ref| max {float(ref.abs().max()):.2f})")
    assert err <= ABS_TOL_STRICT
    if hw == 64:
        # anchored to truth, not to another fp32 summation order (round 5): the reference module in float64 through the same loop
        # (oracle/make_golden.py f3_64_f64); the reference's own fp32 run deviates 2.4e-5 from it, so the bound stays north_star's
        g64 = golden("f3_euler50_latent64_f64")
        err64 = float((out.double() - torch.from_numpy(g64["final"])).abs().max())
        bound = max(ABS_TOL_STRICT, 2.0 * float(g64["ref_fp32_max_abs"]))
        print(f"F3 latent64 fp32 vs the float64 run: {err64:.2e} (reference fp32 vs float64: {float(g64['ref_fp32_max_abs']):.2e}; bound {bound:.1e})")
        assert err64 <= bound


@pytest.mark.parametrize("hw", [64, 128])
def test_sdxl_f3_euler50_bf16(gpu, sdxl_bf16, hw):
    ref = torch.from_numpy(golden(f"f3_euler50_latent{hw}")["final"])
    out = _sdxl_loop(sdxl_bf16, torch.bfloat16, gpu, hw, mode="loop")
    err = float((out - ref).abs().max())
    rms = float((out - ref).pow(2).mean().sqrt())
    print(f"F3 latent{hw} bf16: final latent max abs err {err:.2e}, rms {rms:.2e} "
          f"(|ref| max {float(ref.abs().max()):.2f}, rms {float(ref.pow(2).mean().sqrt()):.2f})")
    assert torch.isfinite(out).all()
    mx, rms_b = storage_bound(f"f3_euler50_latent{hw}", "bf16")
    assert rms <= rms_b and err <= mx


@pytest.mark.parametrize("hw", [64, 128])
def test_sdxl_f3_euler50_fp16(gpu, sdxl_fp16, hw):
    ref = torch.from_numpy(golden(f"f3_euler50_latent{hw}")["final"])
    out = _sdxl_loop(sdxl_fp16, torch.float16, gpu, hw, mode="loop")
    err = float((out - ref).abs().max())
    rms = float((out - ref).pow(2).mean().sqrt())
    print(f"F3 latent{hw} fp16: final latent max abs err {err:.2e}, rms {rms:.2e} "
          f"(|ref| max {float(ref.abs().max()):.2f}, rms {float(ref.pow(2).mean().sqrt()):.2f})")
    assert torch.isfinite(out).all()
    mx, rms_b = storage_bound(f"f3_euler50_latent{hw}", "fp16")
    assert rms <= rms_b and err <= mx


# ---------------------------------------------------------------------------------- BASELINE config #3: batch > 1 on SDXL-base
def _sdxl_step_b4(gm, dtype, dev, t):
    x = synth.denoise_inputs(4, 64, 1234)
    xg = {k: v.to(dev, dtype) for k, v in x.items()}
    with torch.no_grad():
        return gm(xg["latent"], t.to(dev), xg["encoder_hidden_states"],
                  {"text_embeds": xg["text_embeds"], "time_ids": xg["time_ids"]})[0].float().cpu()


def test_sdxl_f1_b4_step_fp32(gpu, sdxl_fp32):
    """bs=4, 77-token text conditioning distinct per row, vs the reference's own batch-4 output; scalar and per-row timesteps."""
    g = golden("f1_unet_step_latent64_b4")
    for key, t in (("out", torch.tensor(float(g["timestep"]))), ("out_tvec", torch.from_numpy(g["timesteps_vec"]))):
        ref = torch.from_numpy(g[key])
        out = _sdxl_step_b4(sdxl_fp32, torch.float32, gpu, t)
        err = float((out - ref).abs().max())
        print(f"F1-b4 {key} fp32: max abs err {err:.2e} (|ref| max {float(ref.abs().max()):.2f})")
        assert err <= ABS_TOL_STRICT


def test_sdxl_f1_b4_step_bf16(gpu, sdxl_bf16):
    g = golden("f1_unet_step_latent64_b4")
    ref = torch.from_numpy(g["out"])
    out = _sdxl_step_b4(sdxl_bf16, torch.bfloat16, gpu, torch.tensor(float(g["timestep"])))
    err, rms = float((out - ref).abs().max()), float((out - ref).pow(2).mean().sqrt())
    print(f"F1-b4 bf16: max abs err {err:.2e}, rms err {rms:.2e} (|ref| max {float(ref.abs().max()):.2f}, rms {float(ref.pow(2).mean().sqrt()):.2f})")
    mx, rms_b = storage_bound("f1_unet_step_latent64", "bf16")       # (the rows of a batch are independent: the bs=1 storage error, per row)
    assert err <= mx and rms <= rms_b


def _sdxl_loop_batch(gm, dtype, dev, hw, batch, mode, steps=None):
    x = synth.denoise_inputs(batch, hw, 1234)
    loop = DenoiseLoop(gm, batch, hw, dtype, dev, euler_discrete_tables(50), mode=mode)
    loop.set_conditioning(x["encoder_hidden_states"].to(dev, dtype), x["text_embeds"].to(dev, dtype), x["time_ids"].to(dev, dtype))
    with torch.no_grad():
        if steps is None:
            return loop.denoise(x["latent"]).cpu()
        loop.set_noise(x["latent"])
        loop.run_steps(steps)
        return loop.latent.contiguous(memory_format=torch.contiguous_format).clone().cpu()


def test_sdxl_f3_b2_euler50_fp32(gpu, sdxl_fp32):
    """Two prompts in one batch through the captured loop vs the reference UNet in the same loop (batch 2)."""
    ref = torch.from_numpy(golden("f3_euler50_latent64_b2")["final"])
    out = _sdxl_loop_batch(sdxl_fp32, torch.float32, gpu, 64, 2, "step")
    err = float((out - ref).abs().max())
    print(f"F3-b2 latent64 fp32: max abs err on final latents {err:.2e} (|ref| max {float(ref.abs().max()):.2f})")
    assert err <= ABS_TOL_STRICT


def test_sdxl_f3_b2_euler50_bf16(gpu, sdxl_bf16):
    ref = torch.from_numpy(golden("f3_euler50_latent64_b2")["final"])
    out = _sdxl_loop_batch(sdxl_bf16, torch.bfloat16, gpu, 64, 2, "loop")
    rms, ref_rms = float((out - ref).pow(2).mean().sqrt()), float(ref.pow(2).mean().sqrt())
    print(f"F3-b2 latent64 bf16: final latent rms err {rms:.2e} = {100 * rms / ref_rms:.2f} % of {ref_rms:.2f}")
    assert torch.isfinite(out).all() and rms <= storage_bound("f3_euler50_latent64", "bf16")[1]      # (per row: the bs=1 storage error)


def _sdxl_step_b4_128(gm, dtype, dev):
    x = synth.denoise_inputs(4, 128, 1234)
    cond = {"text_embeds": x["text_embeds"].to(dev, dtype), "time_ids": x["time_ids"].to(dev, dtype)}
    with torch.no_grad():
        return gm(x["latent"].to(dev, dtype), torch.tensor(500.0, device=dev), x["encoder_hidden_states"].to(dev, dtype), cond)[0].float().cpu()


def test_sdxl_f1_b4_latent128_against_the_reference(gpu, sdxl_fp32, sdxl_bf16):
    """BASELINE config #3 at the size it is benchmarked at - bs=4, latent 128 (1024 x 1024), distinct 77-token conditioning per row -
    against the REFERENCE's own output of that step (round 5: `oracle/make_golden.py f1_b4_128`, every 31st value kept).  Before
    this the full-size batch was only compared with its own bs=1 runs."""
    g = golden("f1_unet_step_latent128_b4")
    ref = torch.from_numpy(g["out"])
    out = _sub(_sdxl_step_b4_128(sdxl_fp32, torch.float32, gpu)).reshape(ref.shape)
    err = float((out - ref).abs().max())
    print(f"F1-b4 latent128 fp32: max abs err {err:.2e} (|ref| max {float(g['out_max_abs']):.2f})")
    assert err <= ABS_TOL_STRICT
    out = _sub(_sdxl_step_b4_128(sdxl_bf16, torch.bfloat16, gpu)).reshape(ref.shape)
    err, rms = float((out - ref).abs().max()), float((out - ref).pow(2).mean().sqrt())
    print(f"F1-b4 latent128 bf16: max abs err {err:.2e}, rms err {rms:.2e} (|ref| rms {float(g['out_rms']):.2f})")
    bound_max, bound_rms = storage_bound("f1_unet_step_latent64", "bf16")      # (one step: what bf16 storage alone costs, measured at latent 64)
    assert rms <= bound_rms * float(g["out_rms"]) / 0.45 and err <= 2.0 * bound_max


def test_sdxl_f1_rectangular_latent_against_the_reference(gpu, sdxl_fp32, sdxl_bf16):
    """A RECTANGULAR latent - 96 x 64 (768 x 512 px), the shape class of SDXL's aspect buckets - against the reference's own output
    of that step (`oracle/make_golden.py f1_rect`, every 31st value kept): different row lengths at every level (64 / 32 / 16
    pixels wide, 6,144 / 1,536 tokens) than any square case; then through DenoiseLoop as a captured graph (finite, replay-identical),
    and a size the reference itself cannot run (a side not a multiple of 4) fails loudly."""
    g = golden("f1_unet_step_latent96x64")
    ref = torch.from_numpy(g["out"])
    hw = (int(g["latent_h"]), int(g["latent_w"]))
    x = synth.denoise_inputs(1, hw, 1234)

    def step(gm, dtype):
        cond = {"text_embeds": x["text_embeds"].to(gpu, dtype), "time_ids": x["time_ids"].to(gpu, dtype)}
        with torch.no_grad():
            return gm(x["latent"].to(gpu, dtype), torch.tensor(float(g["timestep"]), device=gpu), x["encoder_hidden_states"].to(gpu, dtype), cond)[0].float().cpu()

    out = _sub(step(sdxl_fp32, torch.float32)).reshape(ref.shape)
    err = float((out - ref).abs().max())
    print(f"F1 96x64 fp32: max abs err {err:.2e} (|ref| max {float(g['out_max_abs']):.2f})")
    assert err <= ABS_TOL_STRICT
    out = _sub(step(sdxl_bf16, torch.bfloat16)).reshape(ref.shape)
    err, rms = float((out - ref).abs().max()), float((out - ref).pow(2).mean().sqrt())
    print(f"F1 96x64 bf16: max abs err {err:.2e}, rms err {rms:.2e} (|ref| rms {float(g['out_rms']):.2f})")
    bound_max, bound_rms = storage_bound("f1_unet_step_latent64", "bf16")      # (one step: what bf16 storage alone costs, measured at latent 64)
    assert rms <= bound_rms * float(g["out_rms"]) / 0.45 and err <= 2.0 * bound_max
    loop = DenoiseLoop(sdxl_bf16, 1, hw, torch.bfloat16, gpu, euler_discrete_tables(4), mode="loop")
    loop.set_conditioning(x["encoder_hidden_states"].to(gpu, torch.bfloat16), x["text_embeds"].to(gpu, torch.bfloat16), x["time_ids"].to(gpu, torch.bfloat16))
    with torch.no_grad():
        a, b = loop.denoise(x["latent"]).cpu(), loop.denoise(x["latent"]).cpu()
    assert a.shape == (1, 4) + hw and torch.isfinite(a).all() and torch.equal(a, b)
    bad = synth.denoise_inputs(1, (30, 64), 1234)
    with pytest.raises(RuntimeError), torch.no_grad():
        sdxl_fp32(bad["latent"].to(gpu), torch.tensor(500.0, device=gpu), bad["encoder_hidden_states"].to(gpu),
                  {"text_embeds": bad["text_embeds"].to(gpu), "time_ids": bad["time_ids"].to(gpu)})


@pytest.mark.parametrize("hw", [64, 128])
def test_sdxl_loop_batch4_rows_are_independent(gpu, sdxl_fp32, hw):
    """DenoiseLoop(batch=4) at BASELINE config #3's size: the samples of a batch never mix, so every row of a batch-4
    run equals the batch-1 run of that prompt (a size-independent property; the bs=1 path is pinned by F1/F3).
    Different M picks different GEMM tiles / K splits, so the comparison is to fp32 summation-order accuracy."""
    steps = 3
    x4 = synth.denoise_inputs(4, hw, 1234)
    out4 = _sdxl_loop_batch(sdxl_fp32, torch.float32, gpu, hw, 4, "step", steps)
    worst = 0.0
    for b in (0, 3):
        loop = DenoiseLoop(sdxl_fp32, 1, hw, torch.float32, gpu, euler_discrete_tables(50), mode="eager")
        loop.set_conditioning(x4["encoder_hidden_states"][b:b + 1].to(gpu), x4["text_embeds"][b:b + 1].to(gpu), x4["time_ids"][b:b + 1].to(gpu))
        with torch.no_grad():
            loop.set_noise(x4["latent"][b:b + 1])
            loop.run_steps(steps)
        worst = max(worst, float((loop.latent.cpu() - out4[b:b + 1]).abs().max()))
    print(f"batch-4 rows vs batch-1 runs, latent {hw}, {steps} steps, fp32: max abs diff {worst:.2e}")
    assert worst <= 2e-4


def test_sdxl_bs4_bf16_loop_graph_1024px(gpu, sdxl_bf16):
    """BASELINE config #3 as benchmarked: bs=4, latent 128, bf16, whole loop in one hipGraph; rows match bs=1 runs."""
    x4 = synth.denoise_inputs(4, 128, 1234)
    out4 = _sdxl_loop_batch(sdxl_bf16, torch.bfloat16, gpu, 128, 4, "loop")
    ref = torch.from_numpy(golden("f3_euler50_latent128")["final"])           # row 0 of the batch is NOT the bs=1 input (different draw)
    assert torch.isfinite(out4).all() and out4.shape == (4, 4, 128, 128)
    loop = DenoiseLoop(sdxl_bf16, 1, 128, torch.bfloat16, gpu, euler_discrete_tables(50), mode="loop")
    loop.set_conditioning(x4["encoder_hidden_states"][1:2].to(gpu, torch.bfloat16), x4["text_embeds"][1:2].to(gpu, torch.bfloat16),
                          x4["time_ids"][1:2].to(gpu, torch.bfloat16))
    with torch.no_grad():
        one = loop.denoise(x4["latent"][1:2]).cpu()
    rms = float((one - out4[1:2]).pow(2).mean().sqrt())
    scale = float(ref.pow(2).mean().sqrt())
    print(f"bs=4 loop graph at 1024 px, bf16: row 1 vs its bs=1 trajectory rms diff {rms:.2e} ({100 * rms / scale:.2f} % of the latent rms)")
    assert rms <= 0.015 * scale


def test_two_loops_interleaved_stay_bit_identical(gpu, sdxl_bf16):
    """Host state is owned per compiled module (ops.ExecContext): a TINY loop and an SDXL loop interleaved in one
    process give bit for bit what each gives alone."""
    tiny_m, tiny = build(TINY, torch.bfloat16, gpu, graph=False)
    tables = euler_discrete_tables(10)
    xt, xtg = tiny_inputs(torch.bfloat16, gpu, 1, 16)
    xs = synth.denoise_inputs(1, 32, 1234)

    def make():
        a = DenoiseLoop(tiny, 1, 16, torch.bfloat16, gpu, tables, cross_dim=TINY.cross_dim, pooled_dim=TINY.pooled_dim, mode="step")
        a.set_conditioning(xtg["encoder_hidden_states"], xtg["text_embeds"], xtg["time_ids"])
        b = DenoiseLoop(sdxl_bf16, 1, 32, torch.bfloat16, gpu, tables, mode="step")
        b.set_conditioning(xs["encoder_hidden_states"].to(gpu, torch.bfloat16), xs["text_embeds"].to(gpu, torch.bfloat16),
                           xs["time_ids"].to(gpu, torch.bfloat16))
        return a, b

    with torch.no_grad():
        a, b = make()
        solo_a = a.denoise(xt["latent"])
        solo_b = b.denoise(xs["latent"])
        a, b = make()
        a.set_noise(xt["latent"]); b.set_noise(xs["latent"])
        sa, sb = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
        a.capture(); b.capture()
        torch.cuda.synchronize()
        for _ in range(10):                        # the two graphs replay concurrently on two streams
            with torch.cuda.stream(sa):
                a.run_steps(1)
            with torch.cuda.stream(sb):
                b.run_steps(1)
        torch.cuda.synchronize()
    assert torch.equal(a.latent.cpu(), solo_a.cpu()) and torch.equal(b.latent.cpu(), solo_b.cpu())
