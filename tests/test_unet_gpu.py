"""End-to-end parity of the compiled UNet and the captured denoise loop.

fp32 ("strict") runs must meet the north_star bound: <= 1e-3 abs on the output /
final latent against the reference's eager path (golden fixtures generated from
the reference itself, oracle/make_golden.py).  bf16 runs use the same kernels
with 8-bit mantissas; their deviation is reported and bounded loosely."""
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


@pytest.mark.parametrize("dtype,tol", [(torch.float32, ABS_TOL_STRICT), (torch.bfloat16, 0.1)])
@pytest.mark.parametrize("batch,hw", [(1, 16), (2, 8), (1, 24)])
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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
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


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
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


# ---------------------------------------------------------------------------------- SDXL-base, reference goldens
@pytest.fixture(scope="module")
def sdxl_fp32(gpu):
    m, gm = build(SDXL_BASE, torch.float32, gpu, graph=False)
    yield gm
    del m, gm
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def sdxl_bf16(gpu):
    m, gm = build(SDXL_BASE, torch.bfloat16, gpu, graph=False)
    yield gm
    del m, gm
    torch.cuda.empty_cache()


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
    assert err <= 0.1          # bf16 storage: ~2 decimal digits through ~600 dependent ops


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


@pytest.mark.parametrize("hw", [64, 128])
def test_sdxl_f3_euler50_bf16(gpu, sdxl_bf16, hw):
    ref = torch.from_numpy(golden(f"f3_euler50_latent{hw}")["final"])
    out = _sdxl_loop(sdxl_bf16, torch.bfloat16, gpu, hw, mode="loop")
    err = float((out - ref).abs().max())
    rms = float((out - ref).pow(2).mean().sqrt())
    print(f"F3 latent{hw} bf16: final latent max abs err {err:.2e}, rms {rms:.2e} "
          f"(|ref| max {float(ref.abs().max()):.2f}, rms {float(ref.pow(2).mean().sqrt()):.2f})")
    assert torch.isfinite(out).all()
    assert rms <= 0.05 * float(ref.pow(2).mean().sqrt())      # bf16 mode: reported, loosely bounded
