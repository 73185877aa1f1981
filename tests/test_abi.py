"""The C-ABI shared library loads and exports every entry point that
include/avc_hip.h declares (no compute: there is no GPU in the CPU test tier),
and the product loader fails loudly instead of falling back."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adaptive_voice_conversion_amd", "csrc")


def test_library_exports_header_symbols():
    so = os.path.join(CSRC, "libavc_hip.so")
    if not os.path.exists(so):
        subprocess.check_call([os.path.join(CSRC, "build.sh")])
    lib = ctypes.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "avc_hip.h")).read()
    names = sorted(set(re.findall(r"\b(avc_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    lib.avc_version.restype = ctypes.c_int
    assert lib.avc_version() >= 100


def test_product_loader_has_no_fallback(monkeypatch, tmp_path):
    from adaptive_voice_conversion_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError, match="build the HIP extension"):
        _lib.load()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "adaptive_voice_conversion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dirpath, f)
