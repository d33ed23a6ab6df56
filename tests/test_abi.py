"""The C-ABI library loads without a GPU and exports exactly what include/*.h declares."""
import os
import re
import subprocess

from stabletriton_amd import _C
from stabletriton_amd.build import lib_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "stabletriton_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(st_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    syms = header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path()], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (st_[a-z0-9_]+)", out)))
    assert [s for s in exported if not s.startswith("st_debug")] == syms, "library exports symbols the header does not declare"


def test_binding_covers_header():
    assert sorted(_C.SIGNATURES) == header_symbols()


def test_abi_version_and_error_string(lib):
    assert lib.st_abi_version() == _C.ABI_VERSION
    # argument validation happens on the host, before any launch: exercise it without a GPU
    rc = lib.st_layer_norm(None, None, None, None, 4, 64, 1e-5, _C.ST_BF16, None)
    assert rc != 0 and b"null" in lib.st_last_error()
    rc = lib.st_attention(1, 1, 1, 1, 1, 8, 8, 2, 40, 80, 80, 80, 80, 1.0, _C.ST_BF16, None)
    assert rc != 0 and b"head_dim" in lib.st_last_error()
    rc = lib.st_linear(16, 16, None, None, None, 16, 4, 8, 12, 12, 8, 0, 0, 0, _C.ST_BF16, None, 0, None, 0, None, None, 0, None, None, 0, None)
    assert rc != 0 and b"multiples" in lib.st_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import pytest
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "lib_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(_C.BackendError, match="no CPU fallback"):
        _C.load()


def test_ops_reject_cpu_tensors():
    import pytest
    import torch
    from stabletriton_amd import ops
    x, w = torch.randn(4, 64), torch.randn(8, 64)
    for call in (lambda: ops.linear(x, w), lambda: ops.layer_norm(x, torch.ones(64), torch.zeros(64), 1e-5),
                 lambda: ops.geglu(x, x), lambda: ops.group_norm(torch.randn(1, 64, 4, 4), 32, torch.ones(64), torch.zeros(64), 1e-5, True),
                 lambda: ops.attention(torch.randn(1, 8, 64), torch.randn(1, 8, 64), torch.randn(1, 8, 64), 1, 0.125)):
        with pytest.raises(ops.BackendError, match="no CPU fallback"):
            call()


def test_library_is_built_from_this_source():
    """Every object was compiled from the sources (and headers, flags) in the tree and the .so links them: a failed compile
    keeps the previous library, and everything measured afterwards would silently be the old code."""
    from stabletriton_amd import build
    assert build.stale_sources() == []
