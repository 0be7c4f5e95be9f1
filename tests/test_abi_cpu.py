"""CPU-side checks of the boundary: the library loads, exports every declared symbol, and
refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import cfrk_amd
    if not os.path.exists(cfrk_amd.library_path()):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cfrk_amd", "csrc"), "-j4"],
                              stdout=subprocess.DEVNULL)
    return cfrk_amd


def test_library_exports_every_declared_symbol(built):
    syms = built.abi_symbols()
    assert len(syms) >= 20
    L = ctypes.CDLL(built.library_path())
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/cfrk_abi.h but not exported"
    assert L.cfrk_abi_version() == 1


def test_strerror_covers_all_codes(built):
    L = built.load_library()
    for code in range(0, -12, -1):
        assert L.cfrk_strerror(code) not in (None, b"", b"unknown error")


def test_no_cpu_fallback(built):
    """without a gfx950 device the product raises; it never routes to a CPU path"""
    n = ctypes.c_int(-1)
    built.load_library().cfrk_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(built.CfrkError) as e:
        built.Context(0)
    assert e.value.code == -8


def test_product_never_imports_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "cfrk_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c", "Makefile")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                if "oracle_lib" in text or "liboracle" in text or "cfrk_oracle" in text:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_python_mirror_constants_equal_the_header(built):
    """every CFRK_* flag / debug bit / parameter index that cfrk_amd/lib.py mirrors has the value include/cfrk_abi.h
    defines (the header is the contract; the round-3 review found it drifting from the code), and the library reads
    no environment variables"""
    import re
    from cfrk_amd import lib
    text = open(os.path.join(ROOT, "include", "cfrk_abi.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(CFRK_[A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\b", text)}
    mirrored = [n for n in dir(lib) if n.startswith("CFRK_")]
    assert len(mirrored) >= 12
    for n in mirrored:
        assert n in defs, f"{n} is not defined in the header"
        assert getattr(lib, n) == defs[n], n
    for dp, _, files in os.walk(os.path.join(ROOT, "cfrk_amd", "csrc")):
        for f in files:
            if f.endswith((".hip", ".h")):
                assert "getenv" not in open(os.path.join(dp, f), errors="ignore").read(), f
