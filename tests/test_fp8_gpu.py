"""fp8 projection path (BASELINE config #5): OCP e4m3 operands on the fp8 matrix pipe.

Two kinds of check.  (1) The GEMM is exact on what it is given: against an fp32 product of the SAME quantised
operands (dequantised in torch) it must agree to fp32 summation accuracy.  (2) End to end against the oracle's fp32
product of the unquantised operands, with the fp8 tolerance stated here: e4m3 keeps 3 mantissa bits (relative
rounding error up to 2^-4 per element, rms 2^-4/sqrt(3)), both operands are rounded, errors of the K products are
independent, so the relative rms error of an output is about sqrt(2) * 2^-4 / sqrt(3) = 5 %; gated at 7 %.
"""
import pytest
import torch

from stabletriton_amd import ops
from tests.util import rel_err

pytestmark = pytest.mark.gpu
FP8_RMS_TOL = 0.07


def rnd(name, shape):
    from stabletriton_amd import synth
    return synth.normal(name, shape, 4321)


def dq_rows(x: ops.Fp8Rows):
    return x.q.view(torch.float8_e4m3fn).float() * x.scale[:, None]


def dq_weight(wq, ws):
    return wq.view(torch.float8_e4m3fn).float() * ws[:, None]


@pytest.mark.parametrize("M,K", [(7, 128), (64, 640), (1024, 1280), (300, 2560), (33, 5120)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_quantize_rows(gpu, M, K, dtype):
    if dtype == torch.float32 and K > 2560:
        pytest.skip("a wave holds at most 2560 fp32 values of a row (the product path quantises bf16 activations)")
    x = (rnd("q.x", (M, K)) * torch.logspace(-3, 2, M)[:, None]).to(gpu, dtype)
    x[0] = 0                                            # an all-zero row must not divide by zero
    q = ops.quantize_fp8(x)
    assert q.q.dtype == torch.uint8 and q.q.shape == (M, K) and q.scale.shape == (M,)
    xf = x.float()
    amax = xf.abs().amax(1)
    assert torch.allclose(q.scale[1:], amax[1:] / 448.0, rtol=1e-6)
    back = dq_rows(q)
    assert torch.isfinite(back).all()
    # every element within half an e4m3 step of its value (step = 2^-3 relative, or the subnormal step 2^-9 of the row scale)
    err = (back - xf).abs()
    bound = torch.maximum(xf.abs() * 2.0 ** -4, q.scale[:, None] * 2.0 ** -10) * 1.001
    assert bool((err <= bound).all())
    assert torch.allclose(back.abs().amax(1)[1:], amax[1:], rtol=1e-6)          # the row maximum maps to +-448: nothing saturates


@pytest.mark.parametrize("M,C", [(1024, 1280), (4096, 640), (77, 2048), (5, 320)])
def test_layer_norm_quantize(gpu, M, C):
    dtype = torch.bfloat16
    x, g, b = rnd("lq.x", (M, C)).to(gpu, dtype), (rnd("lq.g", (C,)) * 0.1 + 1).to(gpu, dtype), (rnd("lq.b", (C,)) * 0.1).to(gpu, dtype)
    q = ops.quantize_fp8(x, layernorm=(g, b, 1e-5))
    ref = torch.nn.functional.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5)
    back = dq_rows(q)
    rms = float((back - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"layer_norm + e4m3 rows M={M} C={C}: relative rms error {rms:.3e}")
    assert rms <= 2.0 ** -4 / 3 ** 0.5 * 1.1


SHAPES = [(1024, 1280, 1280), (1024, 1280, 3840), (4096, 640, 640), (77, 2048, 1280), (200, 256, 136), (1024, 5120, 1280), (2048, 1280, 5120)]


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("epi", ["bias", "bias+residual", "geglu", "silu"])
def test_linear_fp8(gpu, M, K, N, epi):
    geglu = epi == "geglu"
    rows = 2 * N if geglu else N
    x = rnd("l8.x", (M, K)).to(gpu, torch.bfloat16)
    w = (rnd("l8.w", (rows, K)) * K ** -0.5 * torch.logspace(-1, 1, rows)[:, None]).to(gpu)      # channels of very different magnitude
    b = rnd("l8.b", (rows,)).to(gpu, torch.bfloat16)
    res = rnd("l8.r", (M, N)).to(gpu, torch.bfloat16) if "residual" in epi else None
    wq, ws = ops.quantize_weight_fp8(w)
    xq = ops.quantize_fp8(x)
    out = ops.linear_fp8(xq, wq, ws, b, silu=epi == "silu", geglu=geglu, residual=res).float()

    def finish(y):
        y = y + b.float()
        if geglu:
            a, g = y.chunk(2, -1)
            y = a * torch.nn.functional.gelu(g)
        if epi == "silu":
            y = torch.nn.functional.silu(y)
        return y if res is None else y + res.float()
    exact = finish(dq_rows(xq) @ dq_weight(wq, ws).t())               # same quantised operands, fp32 product
    e1 = rel_err(out, exact)
    true = finish(x.float() @ w.float().t())
    rms = float((out - true).pow(2).mean().sqrt() / true.pow(2).mean().sqrt())
    print(f"linear_fp8 M={M} K={K} N={N} {epi}: vs same operands {e1:.2e} (bf16 output rounding), vs unquantised rms {rms:.3f}")
    assert e1 <= 6e-3                                                  # bf16 rounding of the output only
    assert rms <= FP8_RMS_TOL


def test_linear_fp8_rejects_bad_shapes(gpu):
    x = ops.quantize_fp8(torch.zeros(8, 192, device=gpu, dtype=torch.bfloat16))
    wq, ws = ops.quantize_weight_fp8(torch.ones(16, 192, device=gpu))
    with pytest.raises(ops.BackendError, match="multiple of 128"):
        ops.linear_fp8(x, wq, ws)


# ---------------------------------------------------------------------------------- the compiled UNet in fp8 mode
def test_fp8_mode_claims_the_transformer_projections():
    """Host logic (no launch): the fp8 plan of SDXL-base - the q|k|v and GEGLU projections (140, LayerNorm folded) and the
    feed-forward output projections (70) on the fp8 pipe, fed by 210 e4m3 copies their producers' epilogues leave;
    no LayerNorm and no quantisation launch in the graph."""
    from torch import fx
    from stabletriton_amd.optimization import replace_backend
    from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel
    with torch.device("meta"):
        m = UNet2DConditionModel(SDXL_BASE).to(torch.bfloat16)
    gm = replace_backend(fx.symbolic_trace(m), fp8=True)
    plan = gm.rewrite_stats["fp8_plan"]
    assert plan == {"ln_projections": 140, "ff_out_projections": 70, "emitting_producers": 140, "e4m3_tensors": 210}
    assert gm.rewrite_stats["layer_norm_in_gemm"] == 210
    names = [getattr(n.target, "__name__", "") for n in gm.graph.nodes if n.op == "call_function"]
    assert "layer_norm_wrapper" not in names and "ln_linear_wrapper" not in names
    assert names.count("ln_linear_fp8_wrapper") == 140 and names.count("linear_fp8_residual_wrapper") == 70


def test_fp8_unet_step_vs_oracle(gpu, sdxl_bf16_pair):
    """One SDXL-base step (F1 input) with fp8 projections against the reference's own output.  Round 5: the bound is DERIVED - the
    oracle (pinned to the reference bit for bit) was run with bf16 storage AND the plan's projections on e4m3 operands, in fp32
    arithmetic (`unet_oracle.fp8_plan`, `oracle/make_rounded_golden.py f1_fp8`): that format alone moves the output by 0.259 of its
    rms in this random-weight network (70 layers of ~4 % per projection).  The HIP path may deviate by STORAGE_FACTOR = 1.3 times
    that; it measures 0.262 - the kernels add nothing of their own."""
    from stabletriton_amd import synth
    from stabletriton_amd.optimization import optimize_model
    from tests.util import golden
    model, _ = sdxl_bf16_pair
    gm = optimize_model(model, cuda_graph=False, fp8=True)
    x = synth.denoise_inputs(1, 64, 1234)
    xg = {k: v.to(gpu, torch.bfloat16) for k, v in x.items()}
    with torch.no_grad():
        out = gm(xg["latent"], torch.tensor(999.0, device=gpu), xg["encoder_hidden_states"],
                 {"text_embeds": xg["text_embeds"], "time_ids": xg["time_ids"]})[0].float().cpu()
    ref = torch.from_numpy(golden("f1_unet_step_latent64")["out"])
    rms = float((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"F1 with fp8 projections: relative rms error {rms:.3f}, max abs {float((out - ref).abs().max()):.3f} (|ref| max {float(ref.abs().max()):.2f})")
    emulated = float(golden("f1_unet_step_latent64_fp8plan")["rel_rms"])
    print(f"  (e4m3 operands alone, oracle emulation: {emulated:.3f}; ratio {rms / emulated:.2f})")
    assert torch.isfinite(out).all() and rms <= 1.3 * emulated


# ---------------------------------------------------------------------------------- the fp8 plan: e4m3 copies from epilogues, delayed scales
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,N,res", [(1024, 1280, 1280, True), (4096, 640, 640, True), (256, 256, 3840, False), (100, 128, 136, False)])
def test_epilogue_leaves_e4m3_copy_and_maximum(gpu, dtype, M, K, N, res):
    """st_linear_emit8: the copy is e4m3(clamp(stored value / scale)) bit for bit, the partial slots hold the launch's max |value|,
    and st_fp8_update_scales turns them into margin * amax / 448 and clears them (both epilogue forms: 128 x 64 tiles keep the
    fragment layout, 3840 columns take the staged one)."""
    import torch.nn.functional as F
    with ops.ExecContext(hints=False) as ctx:
        x, w, b = rnd("e8.x", (M, K)).to(gpu, dtype), (rnd("e8.w", (N, K)) * K ** -0.5).to(gpu, dtype), rnd("e8.b", (N,)).to(gpu, dtype)
        r = (rnd("e8.r", (M, N)) * 3).to(gpu, dtype) if res else None
        out, stats, act = ops.linear(x, w, b, residual=r, emit_stats=True, emit_q8=("t", 0))
        plain = ops.linear(x, w, b, residual=r)
        assert torch.equal(out, plain)                                   # the copy does not disturb the output
        sc = ctx.fp8
        inv = float(sc.inv_scale[act.index])
        want = (out.float() * inv).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        assert torch.equal(act.q, want)
        amax = sc.amax[act.index].view(torch.float32).max()
        assert float(amax) == float(out.float().abs().max())
        sc.update()
        torch.cuda.synchronize()
        assert abs(float(sc.scale[act.index]) - ops.FP8_MARGIN * float(amax) / 448) <= 1e-6 * float(amax)
        assert int(sc.amax[act.index].abs().max()) == 0
        assert abs(float(sc.inv_scale[act.index]) * float(sc.scale[act.index]) - 1) < 1e-6


@pytest.mark.parametrize("M,K,N,geglu", [(1024, 1280, 3840, False), (1024, 1280, 5120, True), (4096, 640, 2560, True), (256, 128, 64, False),
                                         (4096, 1280, 3840, False), (4096, 1280, 5120, True)])       # the last two: the eight-phase kernel in e4m3, 256 x 256 tiles
def test_ln_linear_fp8x_against_same_operands(gpu, M, K, N, geglu):
    """The fp8 GEMM with the LayerNorm folded (the q|k|v / GEGLU projections of the plan) is exact on what it is given: against
    fp32 arithmetic on the dequantised e4m3 copy and weights; and within the fp8 tolerance of the unquantised bf16 path."""
    import torch.nn.functional as F
    from oracle import unet_oracle as orc
    dtype = torch.bfloat16
    with ops.ExecContext(hints=False) as ctx:
        x0 = (rnd("l8x.x", (M, K)) * 1.7 + 0.4).to(gpu, dtype)
        g, be = (rnd("l8x.g", (K,)) * 0.2 + 1.0).to(gpu, dtype), (rnd("l8x.b", (K,)) * 0.2).to(gpu, dtype)
        rows = 2 * N if geglu else N
        w, bias = (rnd("l8x.w", (rows, K)) * K ** -0.5).to(gpu, dtype), rnd("l8x.bias", (rows,)).to(gpu, dtype)
        eye = torch.eye(K, device=gpu, dtype=dtype)
        for _ in range(2):                                   # second pass: the scale comes from the first pass's maximum
            xg, stats, act = ops.linear(x0, eye, None, emit_stats=True, emit_q8=("t", 1))
            ctx.fp8.update()
        xg, stats, act = ops.linear(x0, eye, None, emit_stats=True, emit_q8=("t", 1))
        wq, ws, c, d = ops.fold_layer_norm_fp8(g, be, w, bias)
        out = ops.linear_fp8x(act, wq, ws, None, geglu=geglu, ln=(stats, c, d, 1e-5))
        scale = float(ctx.fp8.scale[act.index])
        xdq = act.q.view(torch.float8_e4m3fn).float() * scale
        assert float((xdq - xg.float()).abs().max()) <= 2.0 ** -4 * float(xg.float().abs().max())     # in range: nothing saturated
        mean, var = xg.float().mean(1, keepdim=True), xg.float().var(1, unbiased=False, keepdim=True)
        acc = xdq @ dq_weight(wq, ws).t()
        same = torch.rsqrt(var + 1e-5) * (acc - mean * c[None, :]) + d[None, :]
        ref = F.linear(F.layer_norm(xg.float(), (K,), g.float(), be.float(), 1e-5), w.float(), bias.float())
        if geglu:
            same, ref = orc.geglu(same.cpu()).to(gpu), orc.geglu(ref.cpu()).to(gpu)
        e_same = rel_err(out, same)
        rms = float((out.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print(f"ln_linear_fp8x M={M} K={K} N={N} geglu={geglu}: vs same operands {e_same:.2e}, vs unquantised rms {rms:.3f}")
        assert e_same <= 1.2e-2 and rms <= FP8_RMS_TOL * (1.5 if geglu else 1.0)


def test_fp8_feed_forward_pair_keeps_e4m3_between_the_two_gemms(gpu):
    """GEGLU projection -> feed-forward output projection of the plan: the first leaves ONLY an e4m3 copy, the second consumes it
    and emits x', its row statistics and its own copy."""
    import torch.nn.functional as F
    from oracle import unet_oracle as orc
    M, C = 1024, 1280
    dtype = torch.bfloat16
    with ops.ExecContext(hints=False) as ctx:
        x0 = (rnd("ff8.x", (M, C)) * 1.3).to(gpu, dtype)
        g, be = (rnd("ff8.g", (C,)) * 0.2 + 1.0).to(gpu, dtype), (rnd("ff8.b", (C,)) * 0.2).to(gpu, dtype)
        w1, b1 = (rnd("ff8.w1", (8 * C, C)) * C ** -0.5).to(gpu, dtype), rnd("ff8.b1", (8 * C,)).to(gpu, dtype)
        w2, b2 = (rnd("ff8.w2", (C, 4 * C)) * (4 * C) ** -0.5).to(gpu, dtype), rnd("ff8.b2", (C,)).to(gpu, dtype)
        eye = torch.eye(C, device=gpu, dtype=dtype)
        w1q, w1s, c, d = ops.fold_layer_norm_fp8(g, be, w1, b1)
        w2q, w2s = ops.quantize_weight_fp8(w2)
        for it in range(3):                                  # two passes settle the scales, the third is the one checked
            if it:
                ctx.fp8.update()
            xg, stats, act = ops.linear(x0, eye, None, emit_stats=True, emit_q8=("t", 2))
            h8 = ops.linear_fp8x(act, w1q, w1s, None, geglu=True, ln=(stats, c, d, 1e-5), emit_q8=("t", 3), want_out=False)
            assert isinstance(h8, ops.Fp8Act) and h8.q.shape == (M, 4 * C)
            y, st2, y8 = ops.linear_fp8x(h8, w2q, w2s, b2, residual=xg, emit_stats=True, emit_q8=("t", 4))
        h = orc.geglu(F.linear(F.layer_norm(xg.float(), (C,), g.float(), be.float(), 1e-5), w1.float(), b1.float()).cpu()).to(gpu)
        ref = F.linear(h, w2.float(), b2.float()) + xg.float()
        delta = (y.float() - xg.float()), (ref - xg.float())
        rms = float((delta[0] - delta[1]).pow(2).mean().sqrt() / delta[1].pow(2).mean().sqrt())
        print(f"fp8 feed-forward pair: relative rms error of the block's update {rms:.3f}")
        assert rms <= 0.1
        s = st2.buf.double().sum(1)
        assert torch.allclose(s[:, 0], y.double().sum(1), rtol=1e-4, atol=1e-2)
        assert torch.equal(y8.q, (y.float() * float(ctx.fp8.inv_scale[y8.index])).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8))


def test_fp8_trajectories_do_not_depend_on_what_ran_before(gpu):
    """Delayed scales are state: without a reset the first step of a trajectory would quantise with the maxima of the previous
    trajectory's LAST step (the other end of the sigma schedule).  DenoiseLoop.set_noise restarts them from measuring passes on
    the trajectory's own first evaluation, so identical inputs give identical outputs whatever ran in between."""
    from stabletriton_amd import synth
    from stabletriton_amd.optimization import optimize_model
    from stabletriton_amd.pipeline import DenoiseLoop
    from stabletriton_amd.scheduler import euler_discrete_tables
    from stabletriton_amd.unet import TINY, UNet2DConditionModel
    m = UNet2DConditionModel(TINY).eval().requires_grad_(False).to(gpu, torch.bfloat16)
    synth.fill_module_(m, 0)
    gm = optimize_model(m, cuda_graph=False, fp8=True)
    x = synth.denoise_inputs(1, 16, 1234, cross_dim=TINY.cross_dim, pooled_dim=TINY.pooled_dim)
    xg = {k: v.to(gpu, torch.bfloat16) for k, v in x.items()}
    loop = DenoiseLoop(gm, 1, 16, torch.bfloat16, gpu, euler_discrete_tables(8), cross_dim=TINY.cross_dim, pooled_dim=TINY.pooled_dim, mode="step")
    loop.set_conditioning(xg["encoder_hidden_states"], xg["text_embeds"], xg["time_ids"])
    with torch.no_grad():
        first = loop.denoise(x["latent"])
        again = loop.denoise(x["latent"])                        # straight after a whole trajectory (scales of its last step)
        other = loop.denoise(x["latent"] * 3.0 + 1.0)            # a trajectory with other ranges in between
        third = loop.denoise(x["latent"])
    assert gm.exec_context.fp8 is not None and gm.exec_context.fp8.sites, "the tiny model has no fp8 sites: the test checks nothing"
    assert torch.isfinite(first).all() and not torch.equal(first, other)
    assert torch.equal(first, again) and torch.equal(first, third)


def test_fp8_trajectories_through_the_diffusers_hook(gpu):
    """The same property through `hooks.DiffusersUNet` (round 5): the hook sees one call per step and no loop object -
    it tells a trajectory start by a timestep that is larger than the last call's (or a new prompt) and restarts the delayed
    scales there.  Identical trajectories under ONE prompt give identical results whatever ran in between."""
    from stabletriton_amd import hooks, synth
    from stabletriton_amd.optimization import optimize_model
    from stabletriton_amd.scheduler import euler_discrete_tables
    from stabletriton_amd.unet import TINY, UNet2DConditionModel
    m = UNet2DConditionModel(TINY).eval().requires_grad_(False).to(gpu, torch.bfloat16)
    synth.fill_module_(m, 0)
    gm = optimize_model(m, cuda_graph=False, fp8=True)
    unet = hooks.DiffusersUNet(gm, TINY, torch.bfloat16, cuda_graph=False)
    x = synth.denoise_inputs(1, 16, 1234, cross_dim=TINY.cross_dim, pooled_dim=TINY.pooled_dim)
    ehs = x["encoder_hidden_states"].to(gpu, torch.bfloat16)
    added = {"text_embeds": x["text_embeds"].to(gpu, torch.bfloat16), "time_ids": x["time_ids"].to(gpu, torch.bfloat16)}
    tb = euler_discrete_tables(6)
    ts = torch.tensor(tb.timesteps, device=gpu)

    def trajectory(noise):
        lat = noise.to(gpu, torch.float32) * tb.init_noise_sigma
        for i, t in enumerate(ts):                                   # 0-dim device tensors, as the pipeline passes them
            eps = unet((lat * float(tb.in_scale()[i])).to(torch.bfloat16), t, encoder_hidden_states=ehs, added_cond_kwargs=added)[0]
            lat = lat + eps.float() * float(tb.dsigma()[i])
        return lat.cpu()

    first = trajectory(x["latent"])
    again = trajectory(x["latent"])                                  # the same prompt object: no new context, only the timestep jump tells
    other = trajectory(x["latent"] * 3.0 + 1.0)
    third = trajectory(x["latent"])
    assert gm.exec_context.fp8 is not None and gm.exec_context.fp8.sites
    assert torch.isfinite(first).all() and not torch.equal(first, other)
    assert torch.equal(first, again) and torch.equal(first, third)
