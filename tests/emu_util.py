"""Loads the CPU lane-level simulation of the HIP kernels (tests/emu/libavc_emu.so).

Test infrastructure only: the product loader never touches this library.  The
library is rebuilt from adaptive_voice_conversion_amd/csrc/*.hip when stale.
"""
import ctypes
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adaptive_voice_conversion_amd", "csrc")
SO = os.path.join(ROOT, "tests", "emu", "libavc_emu.so")
_lib = None


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    srcs = [f for f in glob.glob(os.path.join(CSRC, "*")) if os.path.isfile(f)] + glob.glob(os.path.join(ROOT, "tests", "emu", "*.h")) + \
        glob.glob(os.path.join(ROOT, "tests", "emu", "*.cpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(s) > t for s in srcs if not s.endswith(".so"))


def emu_lib():
    global _lib
    if _lib is None:
        if _stale():
            import fcntl
            with open(os.path.join(ROOT, "tests", "emu", ".build.lock"), "w") as lock:   # one builder among concurrent test processes
                fcntl.flock(lock, fcntl.LOCK_EX)
                if _stale():
                    subprocess.check_call([os.path.join(CSRC, "build_emu.sh")])
        _lib = ctypes.CDLL(SO)
    return _lib


def P(t):
    """device-pointer argument for a (CPU) torch tensor or None"""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


L = ctypes.c_long
I = ctypes.c_int


KINDS = ["emu"]
try:
    import pytest as _pytest
    KINDS = ["emu", _pytest.param("gpu", marks=_pytest.mark.gpu)]
except Exception:  # pragma: no cover
    pass


def backend(kind):
    """(declared ctypes lib, torch device): 'emu' = CPU lane-level simulation, 'gpu' = the real gfx950 library."""
    import torch
    from adaptive_voice_conversion_amd import _lib as product
    if kind == "gpu":
        return product.load(), torch.device("cuda", 0)
    return product.declare(emu_lib()), torch.device("cpu")
