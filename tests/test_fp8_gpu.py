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
    """Host logic (no launch): 443 projections of SDXL-base go to the fp8 path, and with them every LayerNorm."""
    from torch import fx
    from stabletriton_amd.optimization import replace_backend
    from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel
    with torch.device("meta"):
        m = UNet2DConditionModel(SDXL_BASE).to(torch.bfloat16)
    gm = replace_backend(fx.symbolic_trace(m), fp8=True)
    assert gm.rewrite_stats["fp8_projections"] == 443 and gm.rewrite_stats["layer_norm_in_gemm"] == 0
    assert not [n for n in gm.graph.nodes if n.op == "call_function" and getattr(n.target, "__name__", "") == "layer_norm_wrapper"]


def test_fp8_unet_step_vs_oracle(gpu, sdxl_bf16_pair):
    """One SDXL-base step (F1 input) with fp8 projections against the reference's own output.  Tolerance: the
    projections carry ~4 % rms error each (above); through 70 transformer layers with residual connections the output
    deviates by a few tens of percent of its rms in this random-weight network - reported, and gated at 1.5 x the measured value to catch
    a broken kernel (bf16 mode: 0.04)."""
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
    assert torch.isfinite(out).all() and rms <= 0.39          # 1.5 x the measured 0.262
