"""Per-operator parity: HIP kernels (through the C ABI) vs the CPU oracle on the
same seeded inputs.  fp32 = strict mode, bf16 / fp16 = fast modes (tolerances in util.py)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as orc
from stabletriton_amd import ops, synth
from tests.util import HALF_DTYPES, assert_close, rounded

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def rnd(name, shape, scale=1.0):
    return synth.normal(name, shape, 7) * scale


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("shape", [(1, 320, 16, 16), (2, 640, 8, 8), (1, 960, 32, 32), (1, 1920, 8, 8), (2, 2560, 4, 4),
                                   (1, 64, 5, 7), (3, 128, 1, 1),
                                   (1, 960, 128, 128), (1, 320, 128, 128)])      # groups of 491,520 / 163,840 elements, 32 work items
def test_group_norm(gpu, dtype, silu, cl, shape):
    x = rnd("gn.x", shape) * 1.5 + 0.7
    w = rnd("gn.w", (shape[1],)) * 0.2 + 1.0
    b = rnd("gn.b", (shape[1],)) * 0.2
    ref = F.group_norm(rounded(x, dtype), 32, rounded(w, dtype), rounded(b, dtype), 1e-5)
    if silu:
        ref = F.silu(ref)
    xg = x.to(gpu, dtype)
    if cl:
        xg = xg.contiguous(memory_format=torch.channels_last)
    out = ops.group_norm(xg, 32, w.to(gpu, dtype), b.to(gpu, dtype), 1e-5, silu)
    assert out.stride() == xg.stride()
    assert_close(out, ref, dtype, "group_norm")


def test_group_norm_3d_and_large_mean(gpu):
    # 3-D input as in the reference's own self-test (kernels/groupnorm.py:163-169)
    x = rnd("gn3.x", (1, 128, 32))
    gn = torch.nn.GroupNorm(32, 128)
    out = ops.group_norm(x.to(gpu), 32, gn.weight.detach().to(gpu), gn.bias.detach().to(gpu), gn.eps, False)
    assert_close(out, gn(x), torch.float32, "group_norm 3d")
    # mean >> std: the Chan-combined partials must not cancel
    x = rnd("gn4.x", (1, 320, 64, 64)) * 0.1 + 30.0
    ref = F.group_norm(x, 32, None, None, 1e-5)
    one, zero = torch.ones(320, device=gpu), torch.zeros(320, device=gpu)
    out = ops.group_norm(x.to(gpu).contiguous(memory_format=torch.channels_last), 32, one, zero, 1e-5, False)
    assert_close(out, ref, torch.float32, "group_norm large mean", factor=5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 64, 640), (2, 77, 1280), (5, 64), (3, 9, 2048)])
def test_layer_norm(gpu, dtype, shape):
    x = rnd("ln.x", shape) * 2 - 0.5
    C = shape[-1]
    w, b = rnd("ln.w", (C,)) * 0.2 + 1.0, rnd("ln.b", (C,)) * 0.2
    ref = F.layer_norm(rounded(x, dtype), (C,), rounded(w, dtype), rounded(b, dtype), 1e-5)
    out = ops.layer_norm(x.to(gpu, dtype), w.to(gpu, dtype), b.to(gpu, dtype), 1e-5)
    assert_close(out, ref, dtype, "layer_norm")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,F_", [(64, 2560), (5, 8), (1024, 5120)])
def test_geglu(gpu, dtype, rows, F_):
    xp = rnd("geglu.x", (rows, 2 * F_)) * 2
    ref = orc.geglu(rounded(xp, dtype))
    xg = xp.to(gpu, dtype)
    a, g = xg.chunk(2, dim=-1)                 # strided halves, no copies
    assert_close(ops.geglu(a, g), ref, dtype, "geglu strided")
    assert_close(ops.geglu(a.contiguous(), g.contiguous()), ref, dtype, "geglu contiguous")


LIN_SHAPES = [(1024, 1280, 1280), (256, 640, 2560), (77, 2048, 640), (1, 1280, 320), (2, 320, 1280), (1000, 64, 200),
              (130, 2816, 1280), (4096, 640, 640), (1024, 5120, 1280), (96, 8192, 384)]      # the last two split K in-launch


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N", LIN_SHAPES)
def test_linear(gpu, dtype, M, K, N):
    x, w, b = rnd("lin.x", (M, K)), rnd("lin.w", (N, K)) * K ** -0.5, rnd("lin.b", (N,))
    xr, wr, br = rounded(x, dtype), rounded(w, dtype), rounded(b, dtype)
    xg, wg, bg = x.to(gpu, dtype), w.to(gpu, dtype), b.to(gpu, dtype)
    assert_close(ops.linear(xg, wg, None), F.linear(xr, wr), dtype, "linear")
    assert_close(ops.linear(xg, wg, bg), F.linear(xr, wr, br), dtype, "linear+bias")
    assert_close(ops.linear(xg, wg, bg, silu=True), F.silu(F.linear(xr, wr, br)), dtype, "linear+bias+silu")
    r = rnd("lin.r", (M, N))
    assert_close(ops.linear(xg, wg, bg, residual=r.to(gpu, dtype)), F.linear(xr, wr, br) + rounded(r, dtype), dtype,
                 "linear+bias+residual")


@pytest.mark.parametrize("dtype", HALF_DTYPES)
@pytest.mark.parametrize("M,K,N,geglu", [(4096, 1280, 3840, False),      # 240 tiles of 256 x 256 (wave tiles 128 x 64: two column pairs)
                                         (1024, 1280, 10240, False),     # 256 tiles of 256 x 160 (wave tiles 64 x 80: pair, single tile, pair)
                                         (8192, 640, 640, False),        # 256 x 160 with a short K (ten K tiles)
                                         (4096, 1280, 5120, True),       # GEGLU on 256 x 256: values and gates of 32 output columns per wave (640 tiles: two full rounds + a column-split remainder)
                                         (2048, 1280, 5120, True),       # 320 tiles: one full round on 256 x 256, the last eight tile columns as a launch of their own
                                         (4096, 1280, 10240, False),     # the same split without GEGLU (640 tiles of 256 x 256)
                                         (1024, 1280, 5120, True)])      # GEGLU on 256 x 160: the half tile whose gates come from lane + 32
def test_linear_eight_phase_register_epilogue(gpu, dtype, M, K, N, geglu):
    """The large Linear problems run on the eight-phase kernel; round 5 stores their results straight from the accumulator
    registers over a permuted staging of W (csrc/epilogue.h, direct epilogue): every feature set with a direct instance, both
    wave-tile shapes, against the fp32 product of the rounded operands; the LayerNorm partials against direct sums of the output."""
    rows = 2 * N if geglu else N
    x, w, b = rnd("l8.x", (M, K)), rnd("l8.w", (rows, K)) * K ** -0.5, rnd("l8.b", (rows,))
    xr, wr, br = rounded(x, dtype), rounded(w, dtype), rounded(b, dtype)
    xg, wg, bg = x.to(gpu, dtype), w.to(gpu, dtype), b.to(gpu, dtype)
    if geglu:
        ref = orc.geglu(F.linear(xr, wr, br))
        assert_close(ops.linear(xg, wg, bg, geglu=True), ref, dtype, "linear+bias+geglu")
        return
    base = F.linear(xr, wr)
    assert_close(ops.linear(xg, wg, None), base, dtype, "linear")
    assert_close(ops.linear(xg, wg, bg), base + br, dtype, "linear+bias")
    r = rnd("l8.r", (M, N))
    rg = r.to(gpu, dtype)
    assert_close(ops.linear(xg, wg, bg, residual=rg), base + br + rounded(r, dtype), dtype, "linear+bias+residual")
    for res in (None, rg):
        out, stats = ops.linear(xg, wg, bg, residual=res, emit_stats=True)
        plain = ops.linear(xg, wg, bg, residual=res)
        # (the same bits where both calls run one launch of the same tiles; with row partials the column split of the 640-tile
        #  shape is off and the two calls sum over K in different tile configurations: equal to rounding there)
        if M * N == 4096 * 10240:
            assert_close(out, plain.float().cpu(), dtype, "row partials vs plain")
        else:
            assert torch.equal(out, plain), "emitting the row partials must not change the output"
        o = out.double().cpu()
        s = stats.buf.double().sum(1).cpu()
        assert torch.allclose(s[:, 0], o.sum(1), rtol=1e-5, atol=1e-3)
        assert torch.allclose(s[:, 1], (o ** 2).sum(1), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", HALF_DTYPES)
def test_next_weights_hints_do_not_change_results(gpu, dtype):
    """Inside a compiled module every GEMM-shaped launch is told which weights the NEXT launch reads and touches them
    (ops.WeightPlan); the operator tests above launch without a plan.  Here a three-launch step (LayerNorm-folded q|k|v on
    256 x 256 tiles, a GEGLU projection on 256 x 160 tiles, a small projection with the fragment epilogue) runs under a plan:
    recording pass, then hinted passes - the same bits as without hints.  (Round 5: the register-destination touches of the
    new register epilogue were overwritten in flight; only the full denoise step, where hints are on, showed it.)"""
    M, K = 4096, 1280
    x = rnd("nw.x", (M, K)) * 1.3 + 0.2
    g, be = rnd("nw.g", (K,)) * 0.2 + 1.0, rnd("nw.b", (K,)) * 0.2
    w1, b1 = rnd("nw.w1", (3840, K)) * K ** -0.5, rnd("nw.b1", (3840,))
    w2, b2 = rnd("nw.w2", (10240, K)) * K ** -0.5, rnd("nw.b2", (10240,))
    w3, b3 = rnd("nw.w3", (K, K)) * K ** -0.5, rnd("nw.b3", (K,))
    xg, stats = ops.linear(x.to(gpu, dtype), torch.eye(K).to(gpu, dtype), None, emit_stats=True)
    wf1, c1, d1 = ops.fold_layer_norm(g.to(gpu, dtype), be.to(gpu, dtype), w1.to(gpu, dtype), b1.to(gpu, dtype))
    wf2, c2, d2 = ops.fold_layer_norm(g.to(gpu, dtype), be.to(gpu, dtype), w2.to(gpu, dtype), b2.to(gpu, dtype))
    w3g, b3g = w3.to(gpu, dtype), b3.to(gpu, dtype)

    def step():
        return (ops.ln_linear(xg, stats, wf1, c1, d1, 1e-5), ops.ln_linear(xg[:1024], stats_small, wf2, c2, d2, 1e-5, geglu=True),
                ops.linear(xg, w3g, b3g, residual=xg))

    xs, stats_small = ops.linear(x[:1024].to(gpu, dtype), torch.eye(K).to(gpu, dtype), None, emit_stats=True)
    plain = step()
    ctx = ops.ExecContext()
    for _ in range(4):                   # pass 1 records the launch order, passes 2.. carry the hints
        with ctx.step():
            hinted = step()
        for a, b in zip(plain, hinted):
            assert torch.equal(a, b)
    assert ctx.plan.state == "replay" and len(ctx.plan.entries) == 3


@pytest.mark.parametrize("dtype", HALF_DTYPES)
def test_next_weights_hints_other_kernel_families(gpu, dtype):
    """The same property for the other launches that take a `next_weights` hint: an in-launch split-K GEMM (last arriver's epilogue),
    a halo conv with a K split, a 1x1 conv on the implicit-GEMM loop, the query projection with the text-context attention in its
    epilogue - hinted passes give the bits of the unhinted ones."""
    x = rnd("nwo.x", (1024, 5120)).to(gpu, dtype)
    w = (rnd("nwo.w", (1280, 5120)) * 5120 ** -0.5).to(gpu, dtype)
    b, r = rnd("nwo.b", (1280,)).to(gpu, dtype), rnd("nwo.r", (1024, 1280)).to(gpu, dtype)
    xc = rnd("nwo.xc", (1, 1280, 32, 32)).to(gpu, dtype).contiguous(memory_format=torch.channels_last)
    wc = (rnd("nwo.wc", (1280, 1280, 3, 3)) * 11520 ** -0.5).to(gpu, dtype).contiguous(memory_format=torch.channels_last)
    w1 = (rnd("nwo.w1", (640, 1280, 1, 1)) * 1280 ** -0.5).to(gpu, dtype).contiguous(memory_format=torch.channels_last)
    bc = rnd("nwo.bc", (1280,)).to(gpu, dtype)
    K = 1280
    xq = rnd("nwo.xq", (1, 1024, K)) * 1.2 + 0.1
    g, be = rnd("nwo.g", (K,)) * 0.2 + 1.0, rnd("nwo.be", (K,)) * 0.2
    wq, bq = rnd("nwo.wq", (K, K)) * K ** -0.5, rnd("nwo.bq", (K,))
    kk, vv = rnd("nwo.k", (1, 77, K)).to(gpu, dtype), rnd("nwo.v", (1, 77, K)).to(gpu, dtype)
    xin, st = ops.linear(xq.to(gpu, dtype), torch.eye(K).to(gpu, dtype), None, emit_stats=True)
    wf, c, d = ops.fold_layer_norm(g.to(gpu, dtype), be.to(gpu, dtype), wq.to(gpu, dtype), bq.to(gpu, dtype))

    def step():
        return (ops.linear(x, w, b, residual=r), ops.conv2d(xc, wc, bc, 1, 1), ops.conv2d(xc, w1, None, 1, 0),
                ops.ln_linear_xattn(xin, st, wf, c, d, 1e-5, kk, vv, K // 64, 0.125))

    plain = step()
    ctx = ops.ExecContext()
    for _ in range(3):
        with ctx.step():
            hinted = step()
        for a, b_ in zip(plain, hinted):
            assert torch.equal(a, b_)
    assert ctx.plan.state == "replay" and len(ctx.plan.entries) == 4


@pytest.mark.parametrize("dtype", HALF_DTYPES)
def test_split_k_is_bit_reproducible(gpu, dtype):
    """The in-launch K split sums its slabs in slice order whichever block finishes last: repeated
    launches (and launches interleaved with other split GEMMs that share the workspace) agree bitwise."""
    x, w, b = rnd("sk.x", (1024, 5120)).to(gpu, dtype), (rnd("sk.w", (1280, 5120)) * 5120 ** -0.5).to(gpu, dtype), rnd("sk.b", (1280,)).to(gpu, dtype)
    x2, w2 = rnd("sk.x2", (77, 2048)).to(gpu, dtype), (rnd("sk.w2", (640, 2048)) * 2048 ** -0.5).to(gpu, dtype)
    first = ops.linear(x, w, b)
    for _ in range(5):
        ops.linear(x2, w2, None)
        assert torch.equal(ops.linear(x, w, b), first)
    xc = rnd("sk.xc", (1, 1280, 32, 32)).to(gpu, dtype).contiguous(memory_format=torch.channels_last)
    wc = (rnd("sk.wc", (1280, 1280, 3, 3)) * 11520 ** -0.5).to(gpu, dtype).contiguous(memory_format=torch.channels_last)
    c0 = ops.conv2d(xc, wc, None, 1, 1)
    for _ in range(3):
        assert torch.equal(ops.conv2d(xc, wc, None, 1, 1), c0)
    ref = F.conv2d(xc.float().cpu(), wc.float().cpu(), None, 1, 1)
    assert_close(c0, ref, dtype, "conv split-K")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,F_", [(1024, 640, 2560), (64, 128, 512), (100, 64, 40)])
def test_linear_geglu(gpu, dtype, M, K, F_):
    x, w, b = rnd("lg.x", (2, M // 2, K)), rnd("lg.w", (2 * F_, K)) * K ** -0.5, rnd("lg.b", (2 * F_,))
    ref = orc.geglu(F.linear(rounded(x, dtype), rounded(w, dtype), rounded(b, dtype)))
    out = ops.linear(x.to(gpu, dtype), w.to(gpu, dtype), b.to(gpu, dtype), geglu=True)
    assert_close(out, ref, dtype, "linear+geglu")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N,geglu", [(1024, 1280, 3840, False), (100, 640, 640, False), (256, 640, 2560, True),
                                         (1024, 1280, 5120, True), (77, 128, 64, False), (4096, 640, 1920, False),
                                         # batch-4 shapes: the 256 x 256 eight-phase kernel (plain and GEGLU; the batch-1 GEGLU shape above takes 256 x 160)
                                         (4096, 1280, 3840, False), (4096, 1280, 5120, True),
                                         # batch 2: 320 tiles of 256 x 256 = one full round + a column-split remainder (round 5)
                                         (2048, 1280, 5120, True)])
def test_ln_linear(gpu, dtype, M, K, N, geglu):
    """LayerNorm folded into the consuming GEMM == LayerNorm followed by Linear (/GEGLU)."""
    x = rnd("lnl.x", (M, K)) * 1.7 + 0.4                  # rows with a non-zero mean
    g, be = rnd("lnl.g", (K,)) * 0.2 + 1.0, rnd("lnl.b", (K,)) * 0.2
    rows = 2 * N if geglu else N
    w, b = rnd("lnl.w", (rows, K)) * K ** -0.5, rnd("lnl.bias", (rows,))
    xr, gr, br, wr, bbr = (rounded(t, dtype) for t in (x, g, be, w, b))
    ref = F.linear(F.layer_norm(xr, (K,), gr, br, 1e-5), wr, bbr)
    if geglu:
        ref = orc.geglu(ref)
    # x itself must come out of a GEMM that emits the row partials: x = x0 @ I + 0 (exact in both dtypes)
    eye = torch.eye(K)
    xg, stats = ops.linear(x.to(gpu, dtype), eye.to(gpu, dtype), None, emit_stats=True)
    if dtype == torch.float32:        # strict mode multiplies split images: 22 significant bits per operand (tests/test_split_gpu.py)
        assert float(((xg.cpu() - xr).abs() / xr.abs().clamp_min(1e-3)).max()) <= 2.0 ** -21
    else:
        assert torch.equal(xg.cpu().float(), xr)
    s = stats.buf.double().sum(1).cpu()
    assert torch.allclose(s[:, 0], xr.double().sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[:, 1], (xr.double() ** 2).sum(1), rtol=1e-5, atol=1e-3)
    wf, c, d = ops.fold_layer_norm(g.to(gpu, dtype), be.to(gpu, dtype), w.to(gpu, dtype), b.to(gpu, dtype))
    out = ops.ln_linear(xg, stats, wf, c, d, 1e-5, geglu=geglu)
    assert_close(out, ref, dtype, "ln_linear", factor=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,S,H", [(1, 256, 256, 10), (2, 128, 77, 5), (1, 1024, 1024, 20), (1, 100, 33, 2),
                                     (1, 64, 1, 1), (1, 4096, 77, 10), (1, 256, 640, 2), (1, 200, 1000, 3),
                                     (1, 4096, 320, 10), (2, 300, 333, 3), (1, 513, 257, 2), (1, 768, 768, 32), (2, 256, 300, 100)])
# S >= 256: attn32i_kernel (three compute waves per block where that fills more CUs; (1, 768, 768, 32) takes four,
# (1, 4096, 320, 10) the seven-wave blocks, (2, 256, 300, 100) the eight-wave blocks: 200 blocks of 256 rows in one round where
# 224-row blocks would need two); S < 256: attn16v2_kernel
def test_attention(gpu, dtype, B, T, S, H):
    C = H * 64
    q, k, v = rnd("att.q", (B, T, C)), rnd("att.k", (B, S, C)), rnd("att.v", (B, S, C))
    ref = orc.attention_core(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), H)
    out = ops.attention(q.to(gpu, dtype), k.to(gpu, dtype), v.to(gpu, dtype), H, 64 ** -0.5)
    assert_close(out, ref, dtype, "attention")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D", [16, 32, 128])
@pytest.mark.parametrize("B,T,S,H", [(1, 256, 256, 4), (2, 100, 77, 3), (1, 200, 1000, 2), (1, 64, 1, 1), (2, 300, 333, 3), (1, 1024, 1024, 5)])
def test_attention_other_head_sizes(gpu, dtype, D, B, T, S, H):
    """The reference's operator takes head_dim 16 / 32 / 64 / 128 (kernels/attention_fa2.py:118-123); SDXL uses 64 only.
    The other three run on attn_anyd_kernel (csrc/attention_anyd.hip), in every dtype, ragged T / S included; the
    scale argument is honoured (the oracle's is head_dim^-1/2)."""
    C = H * D
    q, k, v = rnd("attd.q", (B, T, C)), rnd("attd.k", (B, S, C)), rnd("attd.v", (B, S, C))
    ref = orc.attention_core(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), H)
    out = ops.attention(q.to(gpu, dtype), k.to(gpu, dtype), v.to(gpu, dtype), H, D ** -0.5)
    assert_close(out, ref, dtype, f"attention head_dim {D}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D", [16, 32, 128])
def test_attention_other_head_sizes_lazy_maximum_and_strided_operands(gpu, dtype, D):
    """attn_anyd_kernel: rows whose maximum outruns the lag late, in the masked last tile, rows that start from a very negative
    first tile, large scores; q / k / v are column slices of one fused projection (strided rows), as in a compiled graph."""
    B, T, S, H = 2, 80, 330, 3
    C = H * D
    q, k, v = rnd("attdl.q", (B, T, C)), rnd("attdl.k", (B, S, C)), rnd("attdl.v", (B, S, C))
    amp = (64 / D) ** 0.5                     # the same score scale for every head size
    k[:, 329] = q[:, 2] * 8.0 * amp
    k[:, 200] = q[:, 30] * 5.0 * amp
    k[:, 70] = q[:, 79] * 1.5 * amp
    k[:, :64] = k[:, :64] - q[:, 4:5] * 4.0 * amp
    q[:, 5] = q[:, 5] * 12.0 * amp
    ref = orc.attention_core(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), H)
    qg = torch.cat([q, q], dim=-1).to(gpu, dtype)[..., C:]
    kvg = torch.cat([k, v], dim=-1).to(gpu, dtype)
    out = ops.attention(qg, kvg[..., :C], kvg[..., C:], H, D ** -0.5)
    assert_close(out, ref, dtype, f"attention head_dim {D}, lazy maximum")


def test_attention_rejects_unsupported_head_sizes(gpu):
    for D in (8, 40, 80, 256):
        q = torch.zeros(1, 64, 2 * D, device=gpu, dtype=torch.bfloat16)
        with pytest.raises(ops.BackendError, match="head_dim"):
            ops.attention(q, q, q, 2, D ** -0.5)


def test_attention_peaked_softmax(gpu):
    """Force the online-softmax rescale: one key dominates late in the sequence."""
    B, T, S, H = 1, 128, 320, 2
    q, k, v = rnd("attp.q", (B, T, 128)), rnd("attp.k", (B, S, 128)), rnd("attp.v", (B, S, 128))
    k[:, 300] = q[:, 5] * 6.0
    k[:, 10] = q[:, 70] * 3.0
    for dtype in DTYPES:
        ref = orc.attention_core(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), H)
        out = ops.attention(q.to(gpu, dtype), k.to(gpu, dtype), v.to(gpu, dtype), H, 0.125)
        assert_close(out, ref, dtype, "attention peaked")


@pytest.mark.parametrize("dtype", HALF_DTYPES)
@pytest.mark.parametrize("T,S", [(64, 320), (48, 4096), (16, 77)])
def test_attention_lazy_reference_maximum(gpu, T, S, dtype):
    """The bf16 16-row kernel keeps a row's reference maximum until a tile outruns it by 2^6 (attention.hip ATT_LAG):
    exercise rows whose maximum grows by less than the lag, by more (the exact path), late, and in the masked last
    tile; rows whose first tile holds only very negative scores; rows with very large scores."""
    H = 2
    q, k, v = rnd("attl.q", (1, T, 128)), rnd("attl.k", (1, S, 128)), rnd("attl.v", (1, S, 128))
    last = S - 1
    k[:, 70 % S] = q[:, 1] * 1.5            # second tile (where there is one), growth below the lag
    k[:, last] = q[:, 2] * 8.0              # last key (masked tile when S % 64 != 0), far above the lag
    k[:, S // 2] = q[:, 3] * 3.0
    k[:, :64] = k[:, :64] - q[:, 4:5] * 4.0 * (torch.arange(64)[None, :, None] >= 0)      # row 4: first tile strongly negative
    q[:, 5] = q[:, 5] * 12.0                # row 5: scores of magnitude ~100
    ref = orc.attention_core(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), H)
    out = ops.attention(q.to(gpu, dtype), k.to(gpu, dtype), v.to(gpu, dtype), H, 0.125)
    assert_close(out, ref, dtype, "attention lazy maximum")


@pytest.mark.parametrize("dtype", HALF_DTYPES)
def test_attention_seven_wave_blocks_lazy_maximum(gpu, dtype):
    """attn32i_kernel<7, loader> (more than 128 blocks of 256 rows would be needed: 224-row blocks, T = 512 leaves a ragged
    last block): rows whose maximum outruns the lag in a late tile, in the masked last tile, and rows that start from a
    very negative first tile."""
    B, T, S, H = 9, 512, 330, 8
    q, k, v = rnd("att8.q", (B, T, H * 64)), rnd("att8.k", (B, S, H * 64)), rnd("att8.v", (B, S, H * 64))
    k[:, 329] = q[:, 2] * 8.0
    k[:, 200] = q[:, 300] * 5.0
    k[:, 70] = q[:, 511] * 1.5
    k[:, :64] = k[:, :64] - q[:, 4:5] * 4.0
    ref = orc.attention_core(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), H)
    out = ops.attention(q.to(gpu, dtype), k.to(gpu, dtype), v.to(gpu, dtype), H, 0.125)
    assert_close(out, ref, dtype, "attention, seven-wave blocks")


@pytest.mark.parametrize("dtype", HALF_DTYPES)
@pytest.mark.parametrize("B,T,S,H", [(1, 1024, 1024, 10), (1, 4096, 4096, 10), (2, 1024, 1024, 20), (1, 1024, 77, 20)])
def test_attention_is_bit_stable_when_another_stream_shares_the_cus(gpu, B, T, S, H, dtype):
    """The hardware does not interlock MFMA results against VALU reads and the compiler only protects instructions it
    can see; an inline-asm maximum over fresh accumulators once changed results by 1 ulp whenever the matrix pipe was
    shared with another kernel.  Solo and crowded runs must agree bit for bit."""
    q, k, v = (rnd(f"attc.{n}", (B, L, H * 64)).to(gpu, dtype) for n, L in (("q", T), ("k", S), ("v", S)))
    solo = ops.attention(q, k, v, H, 0.125).clone()
    torch.cuda.synchronize()
    a = torch.randn(2048, 2048, device=gpu, dtype=dtype)
    side = torch.cuda.Stream(device=gpu)
    for _ in range(6):
        with torch.cuda.stream(side):
            for _ in range(8):
                a2 = (a @ a).tanh_()
        out = ops.attention(q, k, v, H, 0.125)
        torch.cuda.synchronize()
        assert torch.equal(out, solo)
    del a2


@pytest.mark.parametrize("B,T,C,H,S", [(1, 1024, 1280, 20, 77), (2, 256, 640, 10, 77), (1, 4096, 640, 10, 77), (1, 128, 128, 2, 5), (3, 384, 192, 3, 200)])
@pytest.mark.parametrize("dtype", HALF_DTYPES)
def test_query_projection_with_text_context_attention_in_its_epilogue(gpu, B, T, C, H, S, dtype):
    """st_ln_linear_xattn == st_ln_linear followed by st_attention, bit for bit (the query tile is rounded to bf16 in LDS
    exactly as the unfused path rounds it in HBM), and both match the oracle."""
    x = rnd("xa.x", (B, T, C)) * 1.3 + 0.2
    g, be = rnd("xa.g", (C,)) * 0.2 + 1.0, rnd("xa.b", (C,)) * 0.2
    w, b = rnd("xa.w", (C, C)) * C ** -0.5, rnd("xa.bias", (C,))
    kv = rnd("xa.kv", (B, S, 2 * C))
    xr, gr, br, wr, bbr, kvr = (rounded(t, dtype) for t in (x, g, be, w, b, kv))
    qref = rounded(F.linear(F.layer_norm(xr, (C,), gr, br, 1e-5), wr, bbr), dtype)
    ref = orc.attention_core(qref, kvr[..., :C], kvr[..., C:], H)
    xg, stats = ops.linear(x.to(gpu, dtype), torch.eye(C).to(gpu, dtype), None, emit_stats=True)
    wf, c, d = ops.fold_layer_norm(g.to(gpu, dtype), be.to(gpu, dtype), w.to(gpu, dtype), b.to(gpu, dtype))
    kvg = kv.to(gpu, dtype)
    kg, vg = kvg[..., :C], kvg[..., C:]                    # strided halves of the fused k|v projection, as in the compiled graph
    q = ops.ln_linear(xg, stats, wf, c, d, 1e-5)
    two = ops.attention(q, kg, vg, H, 0.125)
    one = ops.ln_linear_xattn(xg, stats, wf, c, d, 1e-5, kg, vg, H, 0.125)
    assert torch.equal(one, two)
    assert_close(one, ref, dtype, "query projection + text-context attention", factor=2.0)


CONVS = [  # N, Cin, H, W, Cout, k, stride, pad, upsample
    (1, 320, 32, 32, 320, 3, 1, 1, False), (2, 64, 16, 16, 128, 3, 1, 1, False), (1, 640, 32, 32, 640, 3, 2, 1, False),
    (1, 960, 16, 16, 320, 1, 1, 0, False), (1, 128, 16, 16, 128, 3, 1, 1, True), (1, 4, 32, 32, 320, 3, 1, 1, False),
    (2, 320, 24, 24, 4, 3, 1, 1, False), (1, 192, 9, 7, 64, 3, 1, 1, False), (1, 64, 9, 7, 64, 3, 2, 1, False),
    # the halo loop: 128-, 64- and 32-pixel rows, ragged channel tiles, two images
    (1, 64, 128, 128, 320, 3, 1, 1, False), (1, 128, 64, 64, 200, 3, 1, 1, False), (2, 192, 32, 32, 136, 3, 1, 1, False),
    (1, 64, 32, 32, 160, 3, 1, 1, True), (2, 64, 64, 64, 96, 3, 1, 1, True),
    # 16-pixel rows (the refiner's fourth level: a whole image per tile), two images / a K long enough for a dozen channel slices
    (2, 192, 16, 16, 136, 3, 1, 1, False), (1, 1536, 16, 16, 256, 3, 1, 1, False),
    # 128-pixel rows with 128-channel tiles (the refiner's 384 channels: three tiles, fewer rounds than 160 + 160 + 64)
    (1, 64, 128, 128, 384, 3, 1, 1, False)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv2d(gpu, dtype, cfg):
    N, Cin, H, W, Cout, k, stride, pad, ups = cfg
    x = rnd("conv.x", (N, Cin, H, W))
    w = rnd("conv.w", (Cout, Cin, k, k)) * (Cin * k * k) ** -0.5
    b = rnd("conv.b", (Cout,))
    xr = rounded(x, dtype)
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xr, rounded(w, dtype), rounded(b, dtype), stride=stride, padding=pad)
    cl = torch.channels_last
    xg, wg, bg = x.to(gpu, dtype).contiguous(memory_format=cl), w.to(gpu, dtype).contiguous(memory_format=cl), b.to(gpu, dtype)
    out = ops.conv2d(xg, wg, bg, stride, pad, upsample2x=ups)
    assert out.is_contiguous(memory_format=cl)
    assert_close(out, ref, dtype, "conv2d")
    rb, res = rnd("conv.rb", (N, Cout)), rnd("conv.res", tuple(ref.shape))
    out = ops.conv2d(xg, wg, bg, stride, pad, upsample2x=ups, rowbias=rb.to(gpu, dtype),
                     residual=res.to(gpu, dtype).contiguous(memory_format=cl))
    assert_close(out, ref + rounded(rb, dtype)[:, :, None, None] + rounded(res, dtype), dtype, "conv2d+rowbias+residual")
    # NCHW-contiguous input is accepted too (converted once inside the op)
    assert_close(ops.conv2d(x.to(gpu, dtype), wg, bg, stride, pad, upsample2x=ups), ref, dtype, "conv2d nchw in")


def test_timestep_features(gpu):
    """Integer timesteps (every schedule entry, every size / crop of SDXL's time_ids) take the host's table of the REFERENCE's own
    features: bit for bit the eager path.  The function is ill-conditioned (t * f_j up to 1e3 rad: one ulp of exp() moves a
    feature by 1.2e-4), so anything computed on the device - the non-integer timesteps - is compared at 2e-4."""
    t = torch.tensor([999.0, 500.0, 1.0, 1024.0, 0.0, 981.0, 21.0, 4096.0])
    for dim in (320, 256):
        out = ops.timestep_features(t.to(gpu), dim, torch.float32)
        assert torch.equal(out.cpu(), orc.timestep_features(t, dim))
        for dt in (torch.bfloat16, torch.float16):
            assert torch.equal(ops.timestep_features(t.to(gpu), dim, dt).cpu(), orc.timestep_features(t, dim).to(dt))
        tf = torch.tensor([999.5, 0.25, 6.0, 2.5, 5000.0, -3.0])          # fractional (the refiner's aesthetic score 2.5), beyond the table, negative
        out = ops.timestep_features(tf.to(gpu), dim, torch.float32).cpu()
        ref = orc.timestep_features(tf, dim)
        assert torch.equal(out[2], ref[2])                                  # 6.0 is an integer: the table row
        assert (out - ref).abs().max() < 2e-4 * max(1.0, 5000.0 / 1000.0)
        # the step-indexed form of the denoise loop reads the same rows
        steps = torch.tensor([981.0, 961.0, 941.0], device=gpu)
        idx = torch.tensor([1], dtype=torch.int32, device=gpu)
        assert torch.equal(ops.timestep_features(steps, dim, torch.float32, step=idx, batch=1, t_stride=0).cpu(), orc.timestep_features(torch.tensor([961.0]), dim))


def test_euler_step(gpu):
    from stabletriton_amd.scheduler import euler_discrete_tables
    tb = euler_discrete_tables(50)
    ds, sc = torch.tensor(tb.dsigma()).to(gpu), torch.tensor(tb.in_scale()).to(gpu)
    step = torch.tensor([3], dtype=torch.int32, device=gpu)
    lat = rnd("eu.lat", (1, 4, 16, 16)) * 10
    eps = rnd("eu.eps", (1, 4, 16, 16))
    for dtype in DTYPES:
        lg, eg = lat.to(gpu).clone(), eps.to(gpu, dtype)
        nxt = torch.empty_like(eg)
        ops.euler_step(lg, eg, nxt, ds, sc, step)
        ref = lat + rounded(eps, dtype) * float(tb.dsigma()[3])
        assert (lg.cpu() - ref).abs().max() < 1e-5
        assert_close(nxt, ref * float(tb.in_scale()[4]), dtype, "euler next_in")
    ops.step_advance(step, 50)
    assert int(step.item()) == 4


def test_ops_fail_loudly(gpu):
    x = torch.randn(4, 64)
    w = torch.randn(8, 64)
    with pytest.raises(ops.BackendError):
        ops.linear(x, w)                                   # CPU tensors: no fallback
    with pytest.raises(ops.BackendError):
        ops.linear(x.to(gpu).double(), w.to(gpu).double())   # fp64 is not a supported dtype
    with pytest.raises(ops.BackendError):
        ops.attention(torch.zeros(1, 8, 120, device=gpu), torch.zeros(1, 8, 120, device=gpu),
                      torch.zeros(1, 8, 120, device=gpu), 3, 1.0)   # head_dim 40: not one of the operator's 16 / 32 / 64 / 128
    with pytest.raises(ops.BackendError):                  # the fused query-projection + attention launch takes short contexts only
        C = 128
        xg, st = ops.linear(torch.zeros(1, 128, C, device=gpu, dtype=torch.float16), torch.eye(C, device=gpu, dtype=torch.float16), None, emit_stats=True)
        kv = torch.zeros(1, 300, C, device=gpu, dtype=torch.float16)
        ops.ln_linear_xattn(xg, st, torch.eye(C, device=gpu, dtype=torch.float16), torch.zeros(C, device=gpu), torch.zeros(C, device=gpu), 1e-5, kv, kv, 2, 0.125)


# ---------------------------------------------------------------------------------- GroupNorm statistics from the producer
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,Cin,H,Cout,k,ups", [(1, 320, 32, 320, 3, False), (2, 640, 32, 1280, 3, False), (1, 128, 64, 200, 3, False),
                                                (1, 960, 32, 320, 1, False), (1, 64, 32, 160, 3, True), (1, 320, 64, 640, 3, False)])
def test_conv_column_partials_and_group_norm_from_them(gpu, dtype, N, Cin, H, Cout, k, ups):
    """A conv that also emits per-channel (sum, sum of squares) of what it stored, and the GroupNorm(+SiLU) that runs on
    those partials instead of its own statistics pass - against the oracle's conv -> group_norm."""
    cl = torch.channels_last
    x = rnd("cs.x", (N, Cin, H, H))
    w = rnd("cs.w", (Cout, Cin, k, k), (Cin * k * k) ** -0.5)
    b = rnd("cs.b", (Cout,), 0.5)                         # a visible mean per channel
    g, be = rnd("cs.g", (Cout,)) * 0.2 + 1.0, rnd("cs.be", (Cout,)) * 0.2
    xr = rounded(x, dtype)
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    conv_ref = F.conv2d(xr, rounded(w, dtype), rounded(b, dtype), padding=k // 2)
    out, st = ops.conv2d(x.to(gpu, dtype).contiguous(memory_format=cl), w.to(gpu, dtype).contiguous(memory_format=cl), b.to(gpu, dtype),
                         1, k // 2, upsample2x=ups, emit_colstats=True)
    assert st is not None and st.channels == Cout and (out.shape[2] * out.shape[3]) % st.rows == 0
    tiles = N * out.shape[2] * out.shape[3] // st.rows
    part = st.buf[:tiles].double()
    stored = out.float().permute(0, 2, 3, 1).reshape(tiles, st.rows, Cout).double()
    assert torch.allclose(part[..., 0], stored.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[..., 1], (stored * stored).sum(1), rtol=1e-5, atol=1e-3)
    groups = 32 if Cout % 32 == 0 else 8
    for silu in (False, True):
        y = ops.group_norm_from_stats(out, (st,), groups, g.to(gpu, dtype), be.to(gpu, dtype), 1e-5, silu)
        ref = F.group_norm(conv_ref if dtype == torch.float32 else rounded(conv_ref, dtype), groups, rounded(g, dtype), rounded(be, dtype), 1e-5)
        ref = F.silu(ref) if silu else ref
        assert y.stride() == out.stride()
        assert_close(y, ref, dtype, "group_norm_from_stats", factor=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_group_norm_from_two_sources_unaligned_groups(gpu, dtype):
    """The decoder case (unet_pt.py:352-357): GroupNorm over cat([a, b]) where a comes from a GEMM epilogue (tokens) and b
    from a conv, 1280 + 640 channels in 32 groups of 60 - group 21 straddles the two sources."""
    cl = torch.channels_last
    B, H = 2, 32
    xa, wa, ra = rnd("g2.xa", (B, H * H, 256)), rnd("g2.wa", (1280, 256), 256 ** -0.5), rnd("g2.ra", (B, H * H, 1280))
    a, sa = ops.linear(xa.to(gpu, dtype), wa.to(gpu, dtype), None, residual=ra.to(gpu, dtype), emit_colstats=True)
    xb, wb = rnd("g2.xb", (B, 128, H, H)), rnd("g2.wb", (640, 128, 3, 3), (128 * 9) ** -0.5)
    bb, sb = ops.conv2d(xb.to(gpu, dtype).contiguous(memory_format=cl), wb.to(gpu, dtype).contiguous(memory_format=cl), None, 1, 1,
                        emit_colstats=True)
    assert sa is not None and sb is not None
    a_img = a.reshape(B, H, H, 1280).permute(0, 3, 1, 2)
    cat = torch.cat([a_img, bb], dim=1).contiguous(memory_format=cl)
    g, be = rnd("g2.g", (1920,)) * 0.2 + 1.0, rnd("g2.be", (1920,)) * 0.2
    y = ops.group_norm_from_stats(cat, (sa, sb), 32, g.to(gpu, dtype), be.to(gpu, dtype), 1e-5, True)
    # the same without the concatenated tensor: the apply pass reads the two halves where they lie - bit for bit the same
    y2 = ops.group_norm_from_stats_cat(a_img, bb, (sa, sb), 32, g.to(gpu, dtype), be.to(gpu, dtype), 1e-5, True)
    assert y2.shape == y.shape and y2.is_contiguous(memory_format=cl) and torch.equal(y2, y)
    ref = F.silu(F.group_norm(cat.float().cpu(), 32, rounded(g, dtype), rounded(be, dtype), 1e-5))
    assert_close(y, ref, dtype, "group_norm_from_stats(cat)")
    # a producer that cannot emit (thin conv_in kernel) -> the wrapper falls back to the three-launch GroupNorm
    xi, wi = rnd("g2.xi", (1, 4, 32, 32)), rnd("g2.wi", (320, 4, 3, 3), 36 ** -0.5)
    ci, si = ops.conv2d(xi.to(gpu, dtype).contiguous(memory_format=cl), wi.to(gpu, dtype).contiguous(memory_format=cl), None, 1, 1,
                        emit_colstats=True)
    assert si is None
    y2 = ops.group_norm_from_stats(ci, (si,), 32, g[:320].to(gpu, dtype), be[:320].to(gpu, dtype), 1e-5, False)
    assert_close(y2, F.group_norm(ci.float().cpu(), 32, rounded(g[:320], dtype), rounded(be[:320], dtype), 1e-5), dtype, "fallback")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C0,C1,H,Cout,res,stats", [(1, 1280, 640, 32, 1280, False, True), (2, 640, 320, 64, 640, True, False),
                                                       (1, 320, 320, 128, 320, False, True), (1, 64, 192, 16, 128, True, True)])
def test_conv1x1_over_a_concatenation_that_is_never_written(gpu, dtype, N, C0, C1, H, Cout, res, stats):
    """The resnet shortcut behind a skip connection (unet_pt.py:352-357 -> 74-95): a 1x1 conv of cat([a, b], 1) that reads the
    two tensors where they lie - against the oracle's conv on the concatenation, and bit for bit what ops.conv2d gives on
    the concatenated tensor (same K order, same tiles), statistics included."""
    cl = torch.channels_last
    a, b = rnd("cc.a", (N, C0, H, H)), rnd("cc.b", (N, C1, H, H))
    w, bias = rnd("cc.w", (Cout, C0 + C1, 1, 1), (C0 + C1) ** -0.5), rnd("cc.bias", (Cout,), 0.5)
    r = rnd("cc.r", (N, Cout, H, H)) if res else None
    ag, bg = a.to(gpu, dtype).contiguous(memory_format=cl), b.to(gpu, dtype).contiguous(memory_format=cl)
    wg, biasg = w.to(gpu, dtype).contiguous(memory_format=cl), bias.to(gpu, dtype)
    rg = None if r is None else r.to(gpu, dtype).contiguous(memory_format=cl)
    got = ops.conv2d_cat(ag, bg, wg, biasg, residual=rg, emit_colstats=stats)
    want = ops.conv2d(torch.cat([ag, bg], dim=1), wg, biasg, 1, 0, residual=rg, emit_colstats=stats)
    if stats:
        (got, sg), (want, sw) = got, want
        assert (sg is None) == (sw is None)
        if sg is not None:
            assert sg.rows == sw.rows and sg.channels == sw.channels
            tiles = N * H * H // sg.rows
            assert torch.equal(sg.buf[:tiles], sw.buf[:tiles])
    assert got.is_contiguous(memory_format=cl) and torch.equal(got, want)
    ref = F.conv2d(torch.cat([rounded(a, dtype), rounded(b, dtype)], 1), rounded(w, dtype), rounded(bias, dtype))
    if r is not None:
        ref = ref + rounded(r, dtype)
    assert_close(got, ref, dtype, "conv2d_cat")


def test_conv1x1_cat_falls_back_to_the_concatenation(gpu):
    """Channel counts that are not whole K tiles (or a 3x3 kernel) take torch.cat + conv2d: same result, one more launch."""
    cl = torch.channels_last
    a, b = rnd("cf.a", (1, 48, 16, 16)).to(gpu, torch.bfloat16), rnd("cf.b", (1, 80, 16, 16)).to(gpu, torch.bfloat16)
    w = rnd("cf.w", (64, 128, 1, 1), 128 ** -0.5).to(gpu, torch.bfloat16)
    got = ops.conv2d_cat(a, b, w, None)
    want = ops.conv2d(torch.cat([a, b], dim=1), w, None, 1, 0)
    assert torch.equal(got, want)
