"""The C-ABI library loads and exports every symbol include/snacc_hip.h declares.  No compute
calls here (no GPU on the CPU runner); the product path must fail loudly without a device."""
import ctypes
import re
from pathlib import Path

import pytest

from snacc_amd import hip_backend

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    hip_backend.build()
    return hip_backend.load()


def test_header_symbols_are_exported(lib):
    header = (ROOT / "include" / "snacc_hip.h").read_text()
    declared = set(re.findall(r"\b(snk_[a-z_0-9]+)\s*\(", header))
    assert declared == set(hip_backend.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None, name


def test_version(lib):
    assert lib.snk_version() == hip_backend.ABI_VERSION == 1


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; covered by the gpu tests")
    with pytest.raises(hip_backend.HipBackendError, match="no CPU fallback"):
        hip_backend.HipContext(0)
    from snacc_amd import compressed_size
    p = ROOT / "tests" / "golden" / "tiny.fa"
    with pytest.raises(hip_backend.HipBackendError):
        compressed_size(p, "lz4")


def test_null_ctx_calls_do_not_crash(lib):
    assert lib.snk_pairs(None, 0, 0, None) < 0
    assert lib.snk_set_option(None, b"fast_lanes", 4) < 0
    assert lib.snk_num_sequences(None) < 0
    lib.snk_ctx_destroy(None)
