"""The C-ABI library without a GPU: it builds, loads, exports every symbol include/r50.h declares,
and fails loudly (status + message, no crash, no fallback) when no MI355X is present."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def test_exports_every_declared_symbol(lib_built):
    from implementation_phd_lab_vision_amd import _lib
    header = (ROOT / "include" / "r50.h").read_text()
    declared = set(re.findall(r"\b(r50_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes parsed from include/r50.h"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib_built, name), f"{name} declared in r50.h but not exported by libr50hip.so"
    assert lib_built.r50_version().decode().startswith("r50hip")
    assert lib_built.r50_stem_scratch_bytes(2) == 7 * 64 * 64 + 2 * 230 * 232 * 8


def test_argument_errors_need_no_gpu(lib_built):
    h = C.c_void_p()
    assert lib_built.r50_create(None, 0, 1, 8) == -1
    assert lib_built.r50_create(C.byref(h), 0, 99, 8) == -1 and b"precision" in lib_built.r50_last_error(None)
    assert lib_built.r50_create(C.byref(h), 0, 1, 0) == -1 and b"max_batch" in lib_built.r50_last_error(None)
    assert lib_built.r50_forward(None, None, 1, None, None) == -1
    assert lib_built.r50_set_option(None, b"profile", 1) == -1
    assert lib_built.r50_profile_count(None) == 0
    lib_built.r50_destroy(None)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_fails_loudly_without_a_gpu(lib_built):
    from implementation_phd_lab_vision_amd import _lib
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    h = C.c_void_p()
    assert lib_built.r50_create(C.byref(h), 0, 1, 8) == -2 and h.value is None
    assert b"no HIP device" in lib_built.r50_last_error(None)
    bb = ResNet50Backbone(max_batch=2)
    with pytest.raises(_lib.R50Error):
        bb.to("cpu")                       # the product path has no CPU mode
    with pytest.raises((_lib.R50Error, RuntimeError, AssertionError)):
        bb.to("cuda:0")
    with pytest.raises(_lib.R50Error):
        bb.features(torch.zeros(1, 3, 224, 224))


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import it."""
    pkg = ROOT / "implementation_phd_lab_vision_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        text = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{p} imports the oracle"
        assert "oracle/" not in text or p.suffix != ".py", f"{p} refers to oracle/"
