"""torch.library identity of the operators: Meta kernels trace without a GPU; the HIP kernels are the C ABI."""
import pytest
import torch

import stabletriton_amd.torch_ops  # noqa: F401  (registers torch.ops.st)


def test_meta_kernels_give_shapes_dtypes_and_layouts():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(2, 1024, 1280, dtype=torch.bfloat16, device="cuda")
        w = torch.empty(3840, 1280, dtype=torch.bfloat16, device="cuda")
        y = torch.ops.st.linear_act(x, w, None, False)
        assert y.shape == (2, 1024, 3840) and y.dtype == torch.bfloat16 and y.device.type == "cuda"
        assert torch.ops.st.linear_act(x, w, None, False, True).shape == (2, 1024, 1920)
        q = torch.empty(2, 1024, 1280, dtype=torch.bfloat16, device="cuda")
        kv = torch.empty(2, 77, 1280, dtype=torch.bfloat16, device="cuda")
        assert torch.ops.st.attention(q, kv, kv, 20, 0.125).shape == q.shape
        # every head size the C ABI takes is reachable through the dispatcher op (round 5: the schema carries head_dim)
        for hd in (16, 32, 64, 128):
            assert torch.ops.st.attention(q, kv, kv, 1280 // hd, hd ** -0.5, hd).shape == q.shape
        with pytest.raises(RuntimeError):
            torch.ops.st.attention(q, kv, kv, 20, 0.125, 32)          # 20 heads of 32 are not 1280 channels
        with pytest.raises(RuntimeError):
            torch.ops.st.attention(q, kv, kv, 16, 0.125, 80)          # not a head size of st_attention
        img = torch.empty(2, 320, 64, 64, dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
        g = torch.empty(320, dtype=torch.bfloat16, device="cuda")
        gn = torch.ops.st.group_norm_silu(img, 32, g, g, 1e-5, True)
        assert gn.shape == img.shape and gn.is_contiguous(memory_format=torch.channels_last)
        cw = torch.empty(640, 320, 3, 3, dtype=torch.bfloat16, device="cuda")
        c = torch.ops.st.conv2d_epilogue(img, cw, None, 2, 1)
        assert c.shape == (2, 640, 32, 32) and c.is_contiguous(memory_format=torch.channels_last)
        assert torch.ops.st.conv2d_epilogue(img, cw, None, 1, 1, True).shape == (2, 640, 128, 128)
        assert torch.ops.st.geglu(x, x).shape == x.shape and torch.ops.st.layer_norm(x, w[0], w[0], 1e-5).shape == x.shape


def test_make_fx_traces_the_ops():
    from torch.fx.experimental.proxy_tensor import make_fx

    def f(x, w, b):
        return torch.ops.st.layer_norm(torch.ops.st.linear_act(x, w, b, True), b, b, 1e-5)
    gm = make_fx(f, tracing_mode="fake")(torch.empty(4, 64, device="meta"), torch.empty(64, 64, device="meta"), torch.empty(64, device="meta"))
    names = [str(n.target) for n in gm.graph.nodes if n.op == "call_function"]
    assert names == ["st.linear_act.default", "st.layer_norm.default"]


def test_no_cpu_kernel():
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.st.geglu(torch.zeros(4, 8), torch.zeros(4, 8))


@pytest.mark.gpu
def test_dispatcher_reaches_the_hip_kernels(gpu):
    from stabletriton_amd import ops
    x = torch.randn(64, 256, device=gpu, dtype=torch.bfloat16)
    w = torch.randn(128, 256, device=gpu, dtype=torch.bfloat16)
    b = torch.randn(128, device=gpu, dtype=torch.bfloat16)
    assert torch.equal(torch.ops.st.linear_act(x, w, b, True), ops.linear(x, w, b, silu=True))
    assert torch.equal(torch.ops.st.geglu(x, x), ops.geglu(x, x))
    q = torch.randn(1, 256, 256, device=gpu, dtype=torch.bfloat16)
    kv = torch.randn(1, 77, 256, device=gpu, dtype=torch.bfloat16)
    for hd in (32, 64, 128):
        assert torch.equal(torch.ops.st.attention(q, kv, kv, 256 // hd, hd ** -0.5, hd), ops.attention(q, kv, kv, 256 // hd, hd ** -0.5))
