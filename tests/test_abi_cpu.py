"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/psld_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re

from psld_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "psld_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psld_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"libpsld_hip.so does not export {s}"
        assert s in _lib.SIGNATURES, f"{s} declared in the header but not bound in psld_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in Python but not declared in include/psld_hip.h"
    assert lib.psld_version() == _lib.ABI_VERSION == 8


def test_error_reporting_without_gpu():
    lib = _lib.load()
    # argument validation happens on the host before any launch
    st = lib.psld_axpby_f32(None, 1.0, None, 0.0, None, 4, 0, None)
    assert st != 0 and b"psld_axpby_f32" in lib.psld_last_error()


def test_graft_entry_build_runs():
    """The driver's build check: make (a no-op when the objects are current) + import + symbol binding."""
    import __graft_entry__ as g
    g.build()
