"""The C-ABI library loads and exports every symbol include/gmat_hip.h declares (no GPU needed)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gmat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = text.replace("#define GMAT_API", "")
    return sorted(set(re.findall(r"GMAT_API[^;(]*?\b(\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from gmat_amd.lib import ABI_SYMBOLS
    assert sorted(ABI_SYMBOLS) == _declared_symbols()


def test_product_library_exports_every_declared_symbol():
    import ctypes
    from gmat_amd.lib import lib_path, load
    if not os.path.exists(lib_path()):
        import __graft_entry__ as g
        g.build()
    lib = load()                       # attaches all prototypes -> raises if a symbol is missing
    for name in _declared_symbols():
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.gmat_version()


def test_missing_library_fails_loudly(tmp_path):
    from gmat_amd.lib import load, GmatError
    with pytest.raises(GmatError, match="no CPU fallback"):
        load(str(tmp_path / "libgmat_hip.so"))


def test_product_never_references_the_oracle():
    """No source under gmat_amd/ may include, link or dlopen anything from oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gmat_amd")):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            text = open(os.path.join(dirpath, f), errors="ignore").read()
            if re.search(r"liborc|orc_\w+\(|oracle/|#include\s+\"orc", text):
                bad.append(os.path.join(dirpath, f))
    assert not bad, bad
