"""Drop-in check against the reference's OWN eager module (build container only:
/root/reference is absent on the GPU box, so these tests skip there)."""
import importlib.util
import os

import pytest
import torch
from torch import fx

REF = "/root/reference/src/stabletriton/optimizers/unet_pt.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")


def _ref():
    spec = importlib.util.spec_from_file_location("ref_unet_pt", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_passes_rewrite_the_reference_unet_like_the_reference_does():
    from stabletriton_amd.optimization import replace_backend
    from tests.test_host_logic import EXPECTED
    with torch.device("meta"):
        m = _ref().UNet2DConditionModel()
    gm = replace_backend(fx.symbolic_trace(m))
    for k, v in EXPECTED.items():
        assert gm.rewrite_stats[k] == v, (k, gm.rewrite_stats[k], v)


def test_state_dict_schema_is_identical():
    from stabletriton_amd.unet import SDXL_BASE, UNet2DConditionModel
    with torch.device("meta"):
        a, b = _ref().UNet2DConditionModel(), UNet2DConditionModel(SDXL_BASE)
    sa = {k: tuple(v.shape) for k, v in a.state_dict().items()}
    sb = {k: tuple(v.shape) for k, v in b.state_dict().items()}
    assert sa == sb
