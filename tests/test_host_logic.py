"""Host-side logic that needs no GPU: synthetic data, scheduler tables, fx passes, graph keys."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import fx, nn

from oracle import unet_oracle as orc
from stabletriton_amd import synth
from stabletriton_amd.optimization import replace_backend, _install_context_split
from stabletriton_amd.optimizers import graphs, wrappers
from stabletriton_amd.scheduler import euler_discrete_tables
from stabletriton_amd.unet import SDXL_BASE, TINY, UNet2DConditionModel

# match counts probed on the reference's own UNet
EXPECTED = {"dropout": 227, "attention": 140, "geglu": 70, "linear_silu": 2, "group_norm_silu": 35, "group_norm": 11,
            "layer_norm": 210, "linear": 741, "conv": 51}


def test_synth_is_deterministic_and_named():
    a = synth.param_tensor("conv_in.weight", (320, 4, 3, 3), 0)
    b = synth.param_tensor("conv_in.weight", (320, 4, 3, 3), 0)
    c = synth.param_tensor("conv_out.weight", (320, 4, 3, 3), 0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.std()) * math.sqrt(36) - 1.0) < 0.05          # unit-gain scaling
    # pinned values: any change of the generator invalidates the golden fixtures
    assert [round(float(v), 6) for v in a.flatten()[:3]] == [round(float(v), 6) for v in synth.param_tensor("conv_in.weight", (320, 4, 3, 3), 0).flatten()[:3]]
    u = synth.uniform_pm1("x", 1 << 16, 3)
    assert -1.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean())) < 0.02
    n = synth.normal("n", (1 << 16,), 3)
    assert abs(float(n.std()) - 1.0) < 0.02
    # chunked generation does not change values
    assert torch.equal(synth.uniform_pm1("x", (1 << 20) + 17, 3)[-17:], synth.uniform_pm1("x", 17, 3, offset=1 << 20))


def test_euler_tables_follow_published_formulas():
    t = euler_discrete_tables(50)
    assert t.timesteps[0] == 981.0 and t.timesteps[-1] == 1.0 and len(t.sigmas) == 51 and t.sigmas[-1] == 0.0
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2
    ac = np.cumprod(1 - betas)
    sig981 = math.sqrt((1 - ac[981]) / ac[981])
    assert abs(t.sigmas[0] - sig981) < 1e-5
    assert abs(t.init_noise_sigma - math.sqrt(sig981 ** 2 + 1)) < 1e-5
    assert np.all(np.diff(t.sigmas) < 0)
    # x' = x + eps * dsigma with eps = x / sigma drives x to zero at sigma = 0
    x = orc.euler_denoise(lambda xi, tt: xi * 0 + 1.0, torch.zeros(1, 1), t)
    assert abs(float(x) + float(t.sigmas[0])) < 1e-4


def test_pass_counts_match_reference_probe_on_sdxl():
    with torch.device("meta"):
        m = UNet2DConditionModel(SDXL_BASE)
    assert sum(p.numel() for p in m.parameters()) == 2_567_463_684
    assert len(m.state_dict()) == 1680
    gm = replace_backend(fx.symbolic_trace(m))
    for k, v in EXPECTED.items():
        assert gm.rewrite_stats[k] == v, (k, gm.rewrite_stats[k], v)
    assert gm.rewrite_stats["geglu_in_gemm"] == 70 and gm.rewrite_stats["temb_rowbias"] == 17
    assert gm.rewrite_stats["layer_norm_in_gemm"] == 210 and gm.rewrite_stats["shared_input_gemms"] == 72
    assert gm.rewrite_stats["group_norm_stats"] == 46          # every GroupNorm reads a conv / GEMM output or a cat of two
    assert gm.rewrite_stats["skip_cats_removed"] == 9          # the decoder's nine torch.cat: norm1 and the 1x1 shortcut read the two halves
    assert not any(n.op == "call_function" and n.target is torch.cat and n.kwargs.get("dim", n.args[1] if len(n.args) > 1 else 0) == 1
                   and n.meta.get("skip_cat", True) and len(n.args[0]) == 2 and n.users
                   and all(getattr(u.target, "__name__", "") in ("group_norm_stats_wrapper", "conv2d_wrapper") for u in n.users)
                   for n in gm.graph.nodes)
    assert gm.rewrite_stats["query_projection_in_attention"] == 70     # every cross-attention: to_q's GEMM runs the attention
    _install_context_split(gm)
    assert gm.rewrite_stats["context_outputs"] == 140 and gm.rewrite_stats["time_outputs"] == 1
    # no M=batch GEMM is left in the per-step graph: the whole time path lives in gm.time_module
    assert not [n for n in gm.graph.nodes if n.op == "call_function" and getattr(n.target, "__name__", "") in ("timestep_wrapper", "timestep_embedding_wrapper")]
    assert len([n for n in gm.time_module.graph.nodes if n.op == "call_function"
                and getattr(n.target, "__name__", "") in ("linear_wrapper", "linear_cat_wrapper")]) == 4
    left = [n for n in gm.graph.nodes if n.op == "call_module"]
    assert not [n for n in left if isinstance(gm.get_submodule(n.target), (nn.Linear, nn.Conv2d, nn.GroupNorm, nn.LayerNorm, nn.Dropout))]


def _cpu_backend(monkeypatch):
    """Test-only: stand-in launchers computing with the ORACLE's definitions on the CPU, so the
    rewritten graph can be executed here.  The product has no such path."""
    from stabletriton_amd import ops
    monkeypatch.setattr(ops, "group_norm", lambda x, g, w, b, eps, silu: (F.silu if silu else (lambda t: t))(F.group_norm(x, g, w, b, eps)))
    monkeypatch.setattr(ops, "layer_norm", lambda x, w, b, eps: F.layer_norm(x, w.shape, w, b, eps))
    monkeypatch.setattr(ops, "geglu", lambda s, g: s * F.gelu(g))

    def linear(x, w, b=None, *, silu=False, geglu=False, residual=None, emit_stats=False, emit_colstats=False):
        y = F.linear(x, w, b)
        if silu:
            y = F.silu(y)
        if geglu:
            y = orc.geglu(y)
        y = y if residual is None else y + residual
        if emit_colstats:
            return y, "colstats"
        return (y, "stats") if emit_stats else y
    monkeypatch.setattr(ops, "linear", linear)

    def ln_linear(x, stats, wf, c, d, eps, *, geglu=False, emit_split=False):
        assert stats == "stats"
        mean = x.mean(-1, keepdim=True)
        rstd = (x.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
        y = rstd * (F.linear(x, wf) - mean * c) + d
        return orc.geglu(y) if geglu else y
    monkeypatch.setattr(ops, "ln_linear", ln_linear)
    monkeypatch.setattr(ops, "attention", lambda q, k, v, h, scale: orc.attention_core(q, k, v, h))

    def conv2d(x, w, b, stride, padding, *, upsample2x=False, rowbias=None, residual=None, emit_colstats=False):
        if upsample2x:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        y = F.conv2d(x, w, b, stride=stride, padding=padding)
        if rowbias is not None:
            y = y + rowbias[:, :, None, None]
        y = y if residual is None else y + residual
        return (y, "colstats") if emit_colstats else y
    monkeypatch.setattr(ops, "conv2d", conv2d)

    def group_norm_from_stats(x, sources, g, w, b, eps, silu):
        assert all(s == "colstats" for s in sources) and 1 <= len(sources) <= 2
        return (F.silu if silu else (lambda t: t))(F.group_norm(x, g, w, b, eps))
    monkeypatch.setattr(ops, "group_norm_from_stats", group_norm_from_stats)
    monkeypatch.setattr(ops, "group_norm_from_stats_cat",
                        lambda x0, x1, sources, g, w, b, eps, silu: group_norm_from_stats(torch.cat([x0, x1], 1), sources, g, w, b, eps, silu))
    monkeypatch.setattr(ops, "conv2d_cat", lambda x0, x1, w, b, *, residual=None, emit_colstats=False:
                        conv2d(torch.cat([x0, x1], 1), w, b, 1, 0, residual=residual, emit_colstats=emit_colstats))
    monkeypatch.setattr(ops, "timestep_features", lambda t, dim, dtype, **kw: orc.timestep_features(t, dim).to(dtype))


@pytest.mark.parametrize("fuse", [False, True])
def test_rewritten_graph_is_equivalent_to_eager(monkeypatch, fuse):
    _cpu_backend(monkeypatch)
    torch.manual_seed(0)
    m = UNet2DConditionModel(TINY).eval()
    synth.fill_module_(m, 0)
    x = synth.denoise_inputs(2, 16, 5, cross_dim=TINY.cross_dim, pooled_dim=TINY.pooled_dim)
    cond = {"text_embeds": x["text_embeds"], "time_ids": x["time_ids"]}
    t = torch.tensor(500.0)
    with torch.no_grad():
        ref = m(x["latent"], t, x["encoder_hidden_states"], cond)[0]
        gm = replace_backend(fx.symbolic_trace(m), fuse=fuse)
        if fuse:
            _install_context_split(gm)
        out = gm(x["latent"], t, x["encoder_hidden_states"], cond)[0]
        assert float((out - ref).abs().max()) < 2e-5
        # eager module == oracle restatement (same weights)
        sd = dict(m.state_dict())
        orc_out = orc.unet_forward(sd, x["latent"], t, x["encoder_hidden_states"], x["text_embeds"], x["time_ids"])
        assert float((orc_out - ref).abs().max()) < 2e-5
        if fuse:
            ctx = gm.precompute_context(x["encoder_hidden_states"])
            out2 = gm.forward_with_context(x["latent"], t, ctx, cond)[0]
            assert torch.equal(out, out2)
            # the time path evaluated ahead (one table row per schedule entry) gives the same step
            trow = gm.precompute_time(x["latent"], t, cond)
            assert gm.rewrite_stats["time_outputs"] == len(trow) == 1
            out3 = gm.forward_with_context(x["latent"], None, ctx, None, time_cache=trow)[0]
            assert torch.equal(out, out3)


def test_reference_style_self_tests(monkeypatch):
    """The reference's own pass self-tests (optimizers/*.py __main__ blocks):
    graph code changes and outputs stay within 1e-3."""
    _cpu_backend(monkeypatch)
    from stabletriton_amd.optimizers import (fuse_attention, fuse_geglu, remove_dropout, replace_group_norm,
                                             replace_layer_norm, replace_linear, fuse_timesteps)
    from stabletriton_amd import unet as U

    class Seq(nn.Module):             # remove_dropout.py:8-18
        def __init__(self):
            super().__init__()
            self.lin1, self.lin2, self.lin3 = nn.Linear(5, 5), nn.Linear(5, 5), nn.Linear(5, 5)
            self.nonlin, self.dropout = nn.SiLU(), nn.Dropout(0.0)

        def forward(self, x):
            return self.dropout(self.nonlin(self.lin3(self.lin2(self.lin1(x)))))

    cases = [(Seq(), [remove_dropout, replace_linear], (torch.rand(5, 5),)),
             (U.GEGLU(5, 5), [replace_linear, fuse_geglu], (torch.rand(5, 5),)),
             (nn.Sequential(nn.GroupNorm(32, 128)), [replace_group_norm], (torch.randn(1, 128, 32),)),
             (nn.Sequential(nn.LayerNorm(5)), [replace_layer_norm], (-2.3 + 0.5 * torch.randn(5, 5),)),
             (U.SinusoidalProj(10), [fuse_timesteps], (torch.rand(5),))]

    class OneHead(nn.Module):         # replace_attention.py:139-152 uses Attention(64): one head of 64
        def __init__(self):
            super().__init__()
            self.attn = U.Attention(64, 64)

        def forward(self, x):
            return self.attn(x)
    cases.append((OneHead(), [fuse_attention], (torch.rand(1, 128, 64),)))
    for mod, passes, args in cases:
        mod = mod.eval()
        gm, old = fx.symbolic_trace(mod), fx.symbolic_trace(mod)
        for p in passes:
            p(gm)
        assert gm.code != old.code
        with torch.no_grad():
            assert ((gm(*args) - mod(*args)).abs() < 1e-3).all()


def test_graph_signature_never_contains_tensor_values():
    a, b = torch.tensor(1.0), torch.tensor(2.0)
    assert graphs.signature(((a,), {})) == graphs.signature(((b,), {}))          # reference keys these by value
    assert graphs.signature(((torch.zeros(2, 3),), {})) != graphs.signature(((torch.zeros(3, 2),), {}))
    assert graphs.signature(((1, "x", {"k": a}), {})) == graphs.signature(((1, "x", {"k": b}), {}))


def test_wrappers_are_fx_leaves():
    def f(v, lin):
        return wrappers.linear_wrapper(v, lin, False)
    g = fx.symbolic_trace(f)
    assert any(n.op == "call_function" and n.target is wrappers.linear_wrapper for n in g.graph.nodes)


def test_refiner_spec_and_fp8_pass_counts():
    """SDXL-refiner topology (BASELINE config #5; no reference model, parity unpinned): parameter count of the published
    configuration, pass counts, and the fp8 mode claiming every transformer-block projection."""
    from stabletriton_amd.unet import SDXL_REFINER
    with torch.device("meta"):
        m = UNet2DConditionModel(SDXL_REFINER).to(torch.bfloat16)
    assert sum(p.numel() for p in m.parameters()) == 2_259_526_660
    gm = replace_backend(fx.symbolic_trace(m), fp8=True)
    st = gm.rewrite_stats
    # 4 + 4 layers x 2 resnets on the middle levels down, 3 on the way up, + 4 in the middle block = 8 + 8 + 12 + 12 + 4
    assert st["layer_norm"] == 3 * 44 and st["attention"] == 2 * 44 and st["geglu_in_gemm"] == 44
    assert st["fp8_plan"]["ln_projections"] == 2 * 44 and st["fp8_plan"]["ff_out_projections"] == 44 and st["layer_norm_in_gemm"] == 3 * 44
    left = [n for n in gm.graph.nodes if n.op == "call_module"]
    assert not [n for n in left if isinstance(gm.get_submodule(n.target), (nn.Linear, nn.Conv2d, nn.GroupNorm, nn.LayerNorm, nn.Dropout))]


def test_magic_number_division_is_exact_where_the_kernels_use_it():
    """csrc/gemm_core.h magic_u32 / mg_div: floor(n / d) = (n * (floor(2^32 / d) + 1)) >> 32 whenever n * d < 2^32 - the
    block -> tile map of every GEMM-shaped launch (fill_tile_map falls back to real divisions beyond that)."""
    import random
    rng = random.Random(7)
    for d in list(range(2, 600)) + [rng.randrange(600, 1 << 16) for _ in range(400)]:
        mg = (1 << 32) // d + 1
        assert mg < (1 << 32)
        lim = ((1 << 32) - 1) // d
        for n in [0, 1, d - 1, d, d + 1, 2 * d - 1, lim - 1, lim] + [rng.randrange(0, lim + 1) for _ in range(60)]:
            if 0 <= n <= lim:
                assert (n * mg) >> 32 == n // d, (n, d)
