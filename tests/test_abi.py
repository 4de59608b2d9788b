"""CPU checks of the drop-in boundary: libhmmc_hip.so loads, exports every symbol that
include/hmmc_hip.h declares, and the ctypes signatures in hmmc_amd/_lib.py match the header."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "hmmc_hip.h")
CODE = {"int": "i", "long": "l", "float": "f", "size_t": "z"}


def declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t)\s+(hmmc_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.groups()
        codes = ""
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            if "*" in a or "hmmc_stream_t" in a:
                codes += "p"
            else:
                codes += CODE[a.split()[-2] if len(a.split()) > 1 else a]
        out[name] = (codes, CODE[ret])
    return out


def test_library_builds_and_exports_header_symbols():
    from hmmc_amd import build, _lib
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = declared()
    assert len(decl) >= 10
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in hmmc_hip.h but not exported"


def test_ctypes_signatures_match_header():
    from hmmc_amd import _lib
    decl = declared()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, sig in decl.items():
        assert _lib.SIGNATURES[name] == sig, (name, _lib.SIGNATURES[name], sig)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from hmmc_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        _lib.load()
