"""The strict mode's matrix path: split fp32 operands (csrc/split.h, ST_F32S) on the 16-bit matrix pipe.

north_star's parity bound is the reference's own (optimizers/replace_attention.py:139-152: abs < 1e-3 against eager fp32);
the strict mode meets it with every GEMM-shaped product taken as hi.hi + (hi.lo + lo.hi) * 2^-11 over two IEEE halves per
value.  Checked here: the image is bit-for-bit the definition, the products sit within a small factor of the exact-fp32
kernels' own error against a float64 reference, and values at both ends of the half range behave as documented."""
import pytest
import torch
import torch.nn.functional as F

from stabletriton_amd import ops, synth

pytestmark = pytest.mark.gpu


def rnd(name, shape, scale=1.0):
    return synth.normal(name, shape, 11) * scale


def split_ref(x):
    """the definition, on the CPU: (hi, lo) halves of every value"""
    hi = x.half()
    lo = ((x - hi.float()) * 2048.0).half()
    return hi, lo


@pytest.mark.parametrize("rows,K,scale", [(5, 32, 1.0), (77, 2048, 3.0), (1024, 1280, 1e-3), (3, 64, 1e-6), (16, 96, 2e4)])
def test_split_image_is_the_definition(gpu, rows, K, scale):
    x = rnd("sp.x", (rows, K + 32)) * scale
    xs = ops.split_rows(x.to(gpu)[:, :K]).s          # a row stride that is not K
    img = xs.cpu().view(torch.float16).view(rows, K // 32, 2, 32)
    hi, lo = split_ref(x[:, :K])
    assert torch.equal(img[:, :, 0, :].reshape(rows, K).view(torch.int16), hi.view(torch.int16))
    assert torch.equal(img[:, :, 1, :].reshape(rows, K).view(torch.int16), lo.view(torch.int16))
    back = hi.double() + lo.double() / 2048.0
    normal = x[:, :K].abs() >= 2.0 ** -14
    rel = ((back - x[:, :K].double()).abs() / x[:, :K].abs().double().clamp_min(1e-30))[normal]
    assert rel.numel() == 0 or float(rel.max()) <= 2.0 ** -22, float(rel.max())
    assert float((back - x[:, :K].double()).abs()[~normal].max() if (~normal).any() else 0.0) <= 2.0 ** -35


def _err(out, ref64):
    return float((out.double().cpu() - ref64).abs().max() / ref64.abs().max())


@pytest.mark.parametrize("M,K,N", [(1024, 1280, 1280), (77, 2048, 640), (4096, 640, 640), (1024, 5120, 1280), (2, 320, 1280), (130, 2816, 1280)])
def test_linear_split_against_float64(gpu, M, K, N):
    x, w, b = rnd("spl.x", (M, K)), rnd("spl.w", (N, K)) * K ** -0.5, rnd("spl.b", (N,))
    ref = F.linear(x.double(), w.double(), b.double())
    xg, wg, bg = x.to(gpu), w.to(gpu), b.to(gpu)
    assert ops.STRICT_SPLIT
    e_split = _err(ops.linear(xg, wg, bg), ref)
    ops.STRICT_SPLIT = False
    try:
        e_exact = _err(ops.linear(xg, wg, bg), ref)
    finally:
        ops.STRICT_SPLIT = True
    # 22-bit operands against 24: the rounding of the operands (2^-23 each, independent) adds to the fp32 accumulation error
    assert e_split <= 2e-6, (e_split, e_exact)
    assert e_split <= 8 * e_exact + 2e-7, (e_split, e_exact)


@pytest.mark.parametrize("M,K,N,geglu,res", [(1024, 1280, 5120, True, False),      # the strict GEGLU projection at batch 1: 256 tiles of 256 x 160, one round
                                             (2048, 1280, 5120, False, True),      # plain, bias + residual, 256 tiles
                                             (4096, 640, 2560, True, False),       # the 640 level: 512 tiles = two rounds, ten K tiles
                                             (512, 2560, 10240, False, False)])    # long K: 80 K tiles of 32
def test_linear_split_eight_phase_against_float64(gpu, M, K, N, geglu, res):
    """Round 5: split operands on the eight-phase kernel (256 x 160 tiles, 64 x 80 wave tiles, two accumulator sets; staged fp32
    epilogue) where a launch is whole rounds of such tiles - same error class as the single-phase loop it replaces there, and the
    same bits from launch to launch."""
    rows = 2 * N if geglu else N
    x, w, b = rnd("sp8.x", (M, K)), rnd("sp8.w", (rows, K)) * K ** -0.5, rnd("sp8.b", (rows,))
    r = rnd("sp8.r", (M, N)) if res else None
    ref = F.linear(x.double(), w.double(), b.double())
    if geglu:
        ref = ref[:, :N] * F.gelu(ref[:, N:])
    if res:
        ref = ref + r.double()
    xg, wg, bg = x.to(gpu), w.to(gpu), b.to(gpu)
    out = ops.linear(xg, wg, bg, geglu=geglu, residual=None if r is None else r.to(gpu))
    assert torch.equal(out, ops.linear(xg, wg, bg, geglu=geglu, residual=None if r is None else r.to(gpu)))
    assert _err(out, ref) <= 3e-6, _err(out, ref)


def test_linear_split_is_deterministic_and_matches_operand_rounding(gpu):
    """The product of the split images equals, to accumulation order, the fp64 product of the values the images hold."""
    M, K, N = 256, 640, 320
    x, w = rnd("spd.x", (M, K)) * 4, rnd("spd.w", (N, K)) * K ** -0.5
    xh, xl = split_ref(x)
    wh, wl = split_ref(w)
    xv, wv = xh.double() + xl.double() / 2048, wh.double() + wl.double() / 2048
    ref = xv @ wv.T - (xl.double() / 2048) @ (wl.double() / 2048).T          # the lo.lo term is dropped by design
    out = ops.linear(x.to(gpu), w.to(gpu), None)
    assert torch.equal(out, ops.linear(x.to(gpu), w.to(gpu), None))
    assert _err(out, ref) <= 5e-7


@pytest.mark.parametrize("scale,bound", [(1e-6, 5e-5), (1e-3, 3e-6), (1.0, 3e-6), (3e3, 3e-6)])
def test_split_magnitudes(gpu, scale, bound):
    """Both ends of the half range.  Activations of a few thousand (SDXL's residual stream) are ordinary values.  Below
    2^-14 = 6e-5 the halves are subnormal: such values keep an ABSOLUTE precision of 2^-36, not 22 bits - a tensor that is
    1e-6 everywhere multiplies to 1e-5 relative (invisible next to the O(1) activations the 1e-3 gate is about)."""
    M, K, N = 128, 1280, 256
    x, w = rnd("spm.x", (M, K)) * scale, rnd("spm.w", (N, K)) * K ** -0.5
    ref = x.double() @ w.double().T
    assert _err(ops.linear(x.to(gpu), w.to(gpu), None), ref) <= bound


def test_split_overflow_is_loud(gpu):
    x = torch.full((4, 64), 1.0e5)
    w = torch.ones(8, 64)
    assert not torch.isfinite(ops.linear(x.to(gpu), w.to(gpu), None)).all()      # beyond the half range: inf / NaN, never a saturated product


@pytest.mark.parametrize("cfg", [(1, 320, 32, 32, 320, 3, 1, 1, False), (2, 640, 16, 16, 640, 3, 2, 1, False), (1, 1280, 16, 16, 640, 3, 1, 1, True),
                                 (1, 960, 16, 16, 320, 1, 1, 0, False)])
def test_conv_split_against_float64(gpu, cfg):
    N, Cin, H, W, Cout, k, stride, pad, ups = cfg
    x = rnd("spc.x", (N, Cin, H, W))
    w = rnd("spc.w", (Cout, Cin, k, k)) * (Cin * k * k) ** -0.5
    b = rnd("spc.b", (Cout,))
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin.double(), w.double(), b.double(), stride, pad)
    xg = x.to(gpu).contiguous(memory_format=torch.channels_last)
    wg = w.to(gpu).contiguous(memory_format=torch.channels_last)
    out = ops.conv2d(xg, wg, b.to(gpu), stride, pad, upsample2x=ups)
    assert _err(out, ref) <= 2e-6


@pytest.mark.parametrize("B,T,S,H", [(1, 1024, 1024, 4), (2, 256, 77, 3), (1, 333, 330, 2), (1, 64, 4096, 1), (1, 50, 1, 1)])
def test_attention_fp32_against_float64(gpu, B, T, S, H):
    q, k, v = rnd("spa.q", (B, T, H * 64)), rnd("spa.k", (B, S, H * 64)), rnd("spa.v", (B, S, H * 64))
    qh = q.double().view(B, T, H, 64).transpose(1, 2)
    kh = k.double().view(B, S, H, 64).transpose(1, 2)
    vh = v.double().view(B, S, H, 64).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, dim=-1) @ vh).transpose(1, 2).reshape(B, T, H * 64)
    out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, 0.125)
    assert _err(out, ref) <= 3e-6


def test_attention_fp32_peaked_rows(gpu):
    """Scores far apart (one key dominates, the maximum moves late in the row): the lazy reference maximum's slow path."""
    B, T, S, H = 1, 128, 512, 2
    q, k, v = rnd("spp.q", (B, T, H * 64)) * 6, rnd("spp.k", (B, S, H * 64)) * 6, rnd("spp.v", (B, S, H * 64))
    k[:, -3, :] *= 4.0
    qh = q.double().view(B, T, H, 64).transpose(1, 2)
    kh = k.double().view(B, S, H, 64).transpose(1, 2)
    vh = v.double().view(B, S, H, 64).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, dim=-1) @ vh).transpose(1, 2).reshape(B, T, H * 64)
    out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, 0.125)
    assert _err(out, ref) <= 2e-5          # exponents of a few hundred: the scores' 2^-22 shows in the probabilities


# ------------------------------------------------------------------------------------------ split images from the producers
def _image_of(t2):
    return ops.split_rows(t2.contiguous()).s


def _recent(ctx):
    return ctx.__dict__.get("recent_splits", [])


def test_producers_leave_the_split_image_of_what_they_store(gpu):
    """st_arm_split_output: the image a producer writes beside its fp32 output is bit for bit st_split_f32 of that output
    (GEMM epilogues in both forms, GroupNorm apply, attention), and the consumer's launch finds it instead of splitting."""
    with ops.ExecContext() as ctx:
        M, K, N = 1024, 1280, 1280
        x, w, b, r = rnd("em.x", (M, K)).to(gpu), (rnd("em.w", (N, K)) * K ** -0.5).to(gpu), rnd("em.b", (N,)).to(gpu), rnd("em.r", (M, N)).to(gpu)
        for (mm, kk, nn) in ((M, K, N), (4096, 640, 640), (96, 64, 64)):          # staged epilogue, fragment epilogue, a ragged small one
            xx, ww = rnd("em.x2", (mm, kk)).to(gpu), (rnd("em.w2", (nn, kk)) * kk ** -0.5).to(gpu)
            out, st = ops.linear(xx, ww, None, emit_stats=True)
            ent = _recent(ctx)[-1]
            assert ent[0] is out and torch.equal(ent[4].view(torch.int32), _image_of(out).view(torch.int32))
        out, st = ops.linear(x, w, b, residual=r, emit_stats=True)
        img = _recent(ctx)[-1][4]
        # the LayerNorm-folded GEGLU projection behind it: finds the image, leaves its own for ff.net.2
        g, be = (rnd("em.g", (N,)) * 0.2 + 1).to(gpu), (rnd("em.be", (N,)) * 0.2).to(gpu)
        w1, b1 = (rnd("em.w1", (2 * 2560, N)) * N ** -0.5).to(gpu), rnd("em.b1", (2 * 2560,)).to(gpu)
        wf, c, d = ops.fold_layer_norm(g, be, w1, b1)
        seen = []
        real = ops.split_rows
        ops.split_rows = lambda t, shape=None: (seen.append(tuple(t.shape)), real(t, shape))[1]
        try:
            h = ops.ln_linear(out, st, wf, c, d, 1e-5, geglu=True)
        finally:
            ops.split_rows = real
        assert (M, N) not in seen, f"the consumer split its input although the producer left the image: {seen}"
        ent = _recent(ctx)[-1]
        assert ent[0] is h and torch.equal(ent[4].view(torch.int32), _image_of(h).view(torch.int32))
        ops.EMIT_SPLIT = False
        try:
            out2, st2 = ops.linear(x, w, b, residual=r, emit_stats=True)
            h2 = ops.ln_linear(out2, st2, wf, c, d, 1e-5, geglu=True)
        finally:
            ops.EMIT_SPLIT = True
        assert torch.equal(out, out2) and torch.equal(h, h2)           # same bits with and without the producers' images
        # attention -> output projection
        q, k, v = rnd("em.q", (2, 256, 320)).to(gpu), rnd("em.k", (2, 77, 320)).to(gpu), rnd("em.v", (2, 77, 320)).to(gpu)
        o = ops.attention(q, k, v, 5, 0.125)
        ent = _recent(ctx)[-1]
        assert ent[0] is o and torch.equal(ent[4].view(torch.int32), _image_of(o.view(-1, 320)).view(torch.int32))
        # GroupNorm(+SiLU) on a channels_last tensor -> conv
        xi = rnd("em.xi", (2, 320, 16, 16)).to(gpu).contiguous(memory_format=torch.channels_last)
        gw, gb = (rnd("em.gw", (320,)) * 0.2 + 1).to(gpu), rnd("em.gb", (320,)).to(gpu)
        y = ops.group_norm(xi, 32, gw, gb, 1e-5, True)
        ent = _recent(ctx)[-1]
        assert ent[0] is y and torch.equal(ent[4].view(torch.int32), _image_of(y.permute(0, 2, 3, 1).reshape(-1, 320)).view(torch.int32))
        wc = (rnd("em.wc", (64, 320, 3, 3)) * (320 * 9) ** -0.5).to(gpu).contiguous(memory_format=torch.channels_last)
        seen.clear()
        ops.split_rows = lambda t, shape=None: (seen.append(tuple(t.shape)), real(t, shape))[1]
        try:
            co = ops.conv2d(y, wc, None, 1, 1)
        finally:
            ops.split_rows = real
        assert (2 * 16 * 16, 320) not in seen
        ops.EMIT_SPLIT = False
        try:
            co2 = ops.conv2d(ops.group_norm(xi, 32, gw, gb, 1e-5, True), wc, None, 1, 1)
        finally:
            ops.EMIT_SPLIT = True
        assert torch.equal(co, co2)


@pytest.mark.parametrize("D", [16, 32, 128])
def test_attention_fp32_other_head_sizes_against_float64_and_their_image(gpu, D):
    """attn_anyd_kernel in fp32: split operands like the head_dim-64 kernel (same 3e-6 against float64), and the image it
    leaves for the output projection is st_split_f32 of what it stored."""
    B, T, S, H = 2, 200, 333, 4
    C = H * D
    q, k, v = rnd("spd.q", (B, T, C)), rnd("spd.k", (B, S, C)), rnd("spd.v", (B, S, C))
    qh, kh, vh = (t.double().view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * D ** -0.5, dim=-1) @ vh).transpose(1, 2).reshape(B, T, C)
    with ops.ExecContext() as ctx:
        out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, D ** -0.5)
        assert _err(out, ref) <= 3e-6
        if C % 32 == 0:
            ent = _recent(ctx)[-1]
            assert ent[0] is out and torch.equal(ent[4].view(torch.int32), _image_of(out.view(-1, C)).view(torch.int32))


def test_in_place_update_of_a_producer_output_is_not_missed(gpu):
    """A note on a producer's image is honoured only while the output is what the producer wrote: an in-place update bumps the
    tensor's version counter and the consumer splits the updated values itself."""
    with ops.ExecContext():
        xi = rnd("ip.x", (1, 64, 16, 16)).to(gpu).contiguous(memory_format=torch.channels_last)
        gw, gb = (rnd("ip.gw", (64,)) * 0.2 + 1).to(gpu), rnd("ip.gb", (64,)).to(gpu)
        wc = (rnd("ip.wc", (64, 64, 3, 3)) * (64 * 9) ** -0.5).to(gpu).contiguous(memory_format=torch.channels_last)
        y = ops.group_norm(xi, 32, gw, gb, 1e-5, True)
        y.mul_(3.0)
        got = ops.conv2d(y, wc, None, 1, 1)
        want = ops.conv2d(y.clone(memory_format=torch.preserve_format), wc, None, 1, 1)
        assert torch.equal(got, want)


def test_armed_image_is_never_left_unwritten(gpu):
    """An armed launch that cannot emit is rejected (and disarms): a consumer can never pick up an image nobody wrote."""
    from stabletriton_amd import _C
    lib = _C.load()
    img = torch.empty((64, 64), dtype=torch.float32, device=gpu)
    x, w = rnd("arm.x", (64, 64)).to(gpu, torch.bfloat16), rnd("arm.w", (64, 64)).to(gpu, torch.bfloat16)
    _C.check(lib.st_arm_split_output(img.data_ptr(), 64, 64), "arm")
    with pytest.raises(ops.BackendError, match="cannot emit"):
        ops.linear(x, w, None)                       # a bf16 launch
    assert torch.isfinite(ops.linear(x, w, None).float()).all()          # disarmed: the next launch is an ordinary one
    _C.check(lib.st_arm_split_output(img.data_ptr(), 32, 64), "arm")
    with pytest.raises(ops.BackendError, match="armed split image"):
        ops.linear(x.float(), w.float(), None)       # fp32, but another shape
    # a launch rejected by its own argument checks never looks at the arm: the host disarms, so the image (about to be dropped)
    # cannot be written by whatever emitting launch comes next
    _C.check(lib.st_arm_split_output(img.data_ptr(), 64, 64), "arm")
    img.fill_(7.0)
    with pytest.raises(ops.BackendError):
        _C.check(lib.st_linear(x.float().data_ptr(), w.float().data_ptr(), None, None, None, img.data_ptr(), 64, 64, 0, 64, 64, 0, 0, 0, 0, None, 0,
                               None, 0, None, None, 0, None, None, 0, _C.stream_ptr()), "linear")          # K = 0: rejected up front
    ops.linear(x.float(), w.float(), None)           # same shape as the arm: must not find it
    torch.cuda.synchronize()
    assert bool((img == 7.0).all())


def test_attention_takes_k_and_v_from_the_projection_image(gpu):
    """Self-attention in strict mode: the fused q|k|v projection leaves the split image of its output, the attention launch
    reads its K and V columns from it (st_attention_split) - the same bits as the launch that splits K / V tiles itself."""
    from stabletriton_amd import _C
    B, T, H = 2, 320, 5
    C = H * 64
    with ops.ExecContext() as ctx:
        x0 = rnd("qkv.x", (B, T, C)).to(gpu) * 1.5
        eye = torch.eye(C, device=gpu)
        x, st = ops.linear(x0, eye, None, emit_stats=True)
        g, be = (rnd("qkv.g", (C,)) * 0.2 + 1).to(gpu), (rnd("qkv.be", (C,)) * 0.2).to(gpu)
        w = (rnd("qkv.w", (3 * C, C)) * C ** -0.5).to(gpu)
        wf, c, d = ops.fold_layer_norm(g, be, w, None)
        qkv = ops.ln_linear(x, st, wf, c, d, 1e-5, emit_split=True)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        called = []
        lib = _C.load()
        real = lib.st_attention_split
        lib.st_attention_split = lambda *a: (called.append(1), real(*a))[1]
        try:
            o = ops.attention(q, k, v, H, 0.125)
        finally:
            lib.st_attention_split = real
        assert called, "attention did not find the projection's image"
        ops.EMIT_SPLIT = False
        try:
            o2 = ops.attention(q, k, v, H, 0.125)
        finally:
            ops.EMIT_SPLIT = True
        assert torch.equal(o, o2)
