"""CPU-side checks of the C-ABI library: it builds, loads without a GPU, exports every
symbol include/annchor_hip.h declares, and fails loudly (no fallback) without a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g

    g.build()
    from annchor_amd import _native

    return _native


def _declared():
    hdr = open(os.path.join(ROOT, "include", "annchor_hip.h")).read()
    return sorted(set(re.findall(r"\b(annchor_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree(native):
    assert _declared() == native.exported_symbols()


def test_library_exports_every_declared_symbol(native):
    lib = native.load_library()
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_silent_fallback_without_gpu(native):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NativeError, match="no CPU fallback"):
        native.Engine(0)
    from annchor_amd import Annchor

    with pytest.raises(native.NativeError):
        Annchor(["abc", "abd", "xyz"], "levenshtein")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "annchor_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.replace("no CPU oracle", ""), os.path.join(dirpath, fn)
