import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~30 s on 8 CPU cores")


@pytest.fixture(scope="session")
def lib():
    from stabletriton_amd import _C
    return _C.load()


@pytest.fixture(scope="session")
def gpu():
    import torch
    from stabletriton_amd import build
    assert torch.cuda.is_available(), "GPU tests selected but no GPU is visible"
    stale = build.stale_sources()      # a failed compile leaves the previous .so in place: never test that by accident
    assert not stale, f"libstabletriton_amd.so was not built from this source tree (run python -m stabletriton_amd.build): {stale}"
    return torch.device("cuda:0")


# ---- SDXL-base compiled modules shared by the GPU test files (9.6 GB fp32 + 5.1 GB bf16 of synthetic weights: built once) ----
def _build_sdxl(dtype, dev):
    import torch
    from stabletriton_amd import synth
    from stabletriton_amd.optimization import optimize_model
    from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel
    with torch.device("meta"):
        m = UNet2DConditionModel(SDXL_BASE)
    m = m.to_empty(device=dev).to(dtype).eval().requires_grad_(False)
    synth.fill_module_(m, 0)
    return m, optimize_model(m, cuda_graph=False)


@pytest.fixture(scope="session")
def sdxl_fp32_pair(gpu):
    import torch
    pair = _build_sdxl(torch.float32, gpu)
    yield pair
    del pair
    torch.cuda.empty_cache()


@pytest.fixture(scope="session")
def sdxl_bf16_pair(gpu):
    import torch
    pair = _build_sdxl(torch.bfloat16, gpu)
    yield pair
    del pair
    torch.cuda.empty_cache()


@pytest.fixture(scope="session")
def sdxl_fp16_pair(gpu):
    """What the reference call site builds: the module in half precision (load_sdxl_pipeline.py:17-28)."""
    import torch
    pair = _build_sdxl(torch.float16, gpu)
    yield pair
    del pair
    torch.cuda.empty_cache()


@pytest.fixture(scope="session")
def sdxl_fp16(sdxl_fp16_pair):
    return sdxl_fp16_pair[1]


@pytest.fixture(scope="session")
def sdxl_fp32(sdxl_fp32_pair):
    return sdxl_fp32_pair[1]


@pytest.fixture(scope="session")
def sdxl_bf16(sdxl_bf16_pair):
    return sdxl_bf16_pair[1]
