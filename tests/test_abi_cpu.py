"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/klang_mi355.h declares, and FAILS LOUDLY (no CPU fallback) when no GPU is present."""
import os

import pytest

import klang_amd
from klang_amd._lib import HEADER_PATH, LIB_PATH, declared_symbols


def _ensure_built():
    if not os.path.exists(LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()


def test_library_exports_every_declared_symbol():
    _ensure_built()
    names = declared_symbols()
    assert len(names) >= 30 and "klg_process" in names and "klg_fx_process" in names
    L = klang_amd.lib()                     # raises KlangError if any declared symbol is missing
    for n in names:
        assert hasattr(L, n)
    assert L.klg_version() >= 100


def test_header_cites_reference_interfaces():
    text = open(HEADER_PATH).read()
    for cite in ("klang.h:4830-4858", "klang.h:4423-4427", "klang.h:4430-4434", "klang.h:4708-4716"):
        assert cite in text


def test_no_cpu_fallback_without_gpu():
    _ensure_built()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is exercised on CPU-only hosts")
    with pytest.raises(klang_amd.KlangError, match="no CPU fallback"):
        klang_amd.SynthBank("sub2a", synths=1, notes=4)
    with pytest.raises(klang_amd.KlangError):
        klang_amd.FxBank("pingpong", 4)


def test_product_never_imports_oracle():
    """The product tree must not reference oracle/ (only tests/, smoke() and bench.py's cpu_baseline may)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d, _, files in os.walk(os.path.join(root, "klang_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".cpp", ".sh")):
                text = open(os.path.join(d, f), errors="ignore").read()
                assert "klang_oracle" not in text and "oracle/" not in text.replace("oracle/ref/ref_", "").replace("oracle/...", ""), f"{f} references the oracle"
