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
            if a.startswith("const char*"):
                codes += "s"
            elif "*" in a or "hmmc_stream_t" in a:
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


def test_options_and_release_entry_points(monkeypatch):
    """hmmc_set_option / hmmc_get_option / hmmc_tower_release (no GPU needed: settings and an empty event cache), and the
    environment translation of hmmc_amd/_lib.py - the library itself reads no environment variable."""
    from hmmc_amd import _lib
    lib = _lib.load()
    for key in _lib.ENV_OPTIONS.values():
        before = _lib.get_option(key)
        _lib.set_option(key, True)
        assert _lib.get_option(key) is True
        _lib.set_option(key, False)
        assert _lib.get_option(key) is False
        _lib.set_option(key, before)
    assert lib.hmmc_set_option(b"no_such_option", 1) == -1 and lib.hmmc_get_option(b"no_such_option") == -1
    assert lib.hmmc_set_option(None, 1) == -1
    with pytest.raises(KeyError):
        _lib.set_option("nope", 1)
    assert lib.hmmc_tower_release(None, None) == 0           # nothing cached in a process that never ran a tower backward
    # no getenv in the product sources (the scratch-only HMMC_F32_PICK sits behind #ifdef HMMC_SCRATCH)
    import glob
    for src in glob.glob(os.path.join(ROOT, "hmmc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "hmmc_amd", "csrc", "*.h")):
        text = re.sub(r"#ifdef HMMC_SCRATCH.*?#e(lse|ndif)", "", open(src).read(), flags=re.S)
        assert "getenv" not in text, src
    # a variable that is present - even empty - switches its option on when the library is loaded
    monkeypatch.setenv("HMMC_NO_F32_WAVEK", "")
    monkeypatch.setattr(_lib, "_lib", None)
    try:
        _lib.load()
        assert _lib.get_option("no_f32_wavek") is True
    finally:
        _lib.set_option("no_f32_wavek", False)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from hmmc_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        _lib.load()


def test_committed_traffic_record_matches_the_kernel_source_and_the_step_it_was_taken_on():
    """bench.py reports `roofline.traffic` from the newest profiles/r*_gemm_f16_hbm_traffic.json only if the record was taken on
    this gemm_f16.hip and counts the launches of the steps it ran (round 4 shipped a record that divided by one step too many
    after the profiled command lost its trailing step: the bench line would have said `traffic: null`)."""
    import glob, hashlib, json, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r*_gemm_f16_hbm_traffic.json")))
    assert paths, "no committed PMC record"
    rec = json.load(open(paths[-1]))
    src = open(os.path.join(root, "hmmc_amd", "csrc", "gemm_f16.hip"), "rb").read()
    assert rec["gemm_f16_hip_sha256_16"] == hashlib.sha256(src).hexdigest()[:16], "re-collect: scratch/collect_traffic.sh"
    m_steps, m_warm = re.search(r"--steps (\d+)", rec["command"]), re.search(r"--warmup (\d+)", rec["command"])
    ran = int(m_steps.group(1)) + int(m_warm.group(1)) + (0 if "--roofline-steps 0" in rec["command"] else 1)
    assert rec["steps_profiled"] == ran, (rec["steps_profiled"], ran)
    assert rec["launches"] == rec["launches_per_step"] * rec["steps_profiled"]
