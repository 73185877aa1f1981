"""The C-ABI shared library loads and exports every entry point that
include/avc_hip.h declares (no compute: there is no GPU in the CPU test tier),
and the product loader fails loudly instead of falling back."""
import ctypes
import torch  # noqa: F401  (before the library: one HIP runtime per process, see _lib.load)
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


def test_tuning_is_plan_scoped_or_thread_local_never_process_wide():
    """include/avc_hip.h: "the library has NO process-wide mutable state".  avc_set_tuning edits the CALLING thread's op-level
    copy only; a plan captures its own avc_tuning at creation (host-only calls: no GPU needed)."""
    import threading
    from adaptive_voice_conversion_amd import _lib
    from adaptive_voice_conversion_amd.engine import Plan
    from oracle import avc_oracle as O
    lib = _lib.load()
    base = _lib.Tuning()
    lib.avc_get_op_tuning(ctypes.byref(base))
    assert base.struct_size == ctypes.sizeof(_lib.Tuning) and base.conv_ck5 == 8 and base.wgrad_batch == 12
    assert lib.avc_set_tuning(b"no_such_knob", 1) == -1
    seen = {}

    def other():
        assert lib.avc_set_tuning(b"conv_ck5", 16) == 0 and lib.avc_set_tuning(b"conv_x3", 2) == 0
        t = _lib.Tuning()
        lib.avc_get_op_tuning(ctypes.byref(t))
        seen["other"] = (t.conv_ck5, t.conv_x3)
    th = threading.Thread(target=other)
    th.start()
    th.join()
    mine = _lib.Tuning()
    lib.avc_get_op_tuning(ctypes.byref(mine))
    assert seen["other"] == (16, 2) and (mine.conv_ck5, mine.conv_x3) == (8, 0)      # the other thread's edits stayed there
    cfg = O.tiny_config(c_h=128, c_bank=32)
    plain, x3 = Plan(cfg, 2, 40, lib=lib), Plan(cfg, 2, 40, lib=lib, tuning={"conv_x3": 2})
    assert x3.workspace_floats > plain.workspace_floats                                # the x3 plan carries its own weight images
    assert Plan(cfg, 2, 40, lib=lib).workspace_floats == plain.workspace_floats        # ... and left no trace for the next plan
    with pytest.raises(KeyError):
        Plan(cfg, 2, 40, lib=lib, tuning={"conv_rs": 1})


def test_ragged_griffin_lim_validates_the_host_offsets():
    """avc_dsp_griffin_lim_ragged indexes frames and samples through the DEVICE offset array without bounds checks; the library
    validates the host copy first and refuses (before any launch: no GPU needed) offsets that are not monotone from 0 to Ttot
    (-1) or an utterance too short for the reflect padding of its STFT (-6, as the equal-length entry point)."""
    import numpy as np
    from adaptive_voice_conversion_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(16)   # never dereferenced: every case below is rejected by the host-side checks
    n_fft, hop, win = 2048, 300, 1200

    def call(toff, Ttot, B=None):
        th = np.ascontiguousarray(toff, dtype=np.int32)
        return lib.avc_dsp_griffin_lim_ragged(one, one, ctypes.c_void_p(th.ctypes.data), len(toff) - 1 if B is None else B, Ttot, n_fft, hop, win, 1,
                                              one, one, one, one, None)
    assert call([0, 40, 30, 90], 90) == -1          # not monotone
    assert call([1, 40, 90], 90) == -1              # does not start at 0
    assert call([0, 40, 80], 90) == -1              # does not end at Ttot
    assert call([0, 40, 44, 90], 90) == -6          # 4 frames: 300 * 3 <= 1024
    assert lib.avc_dsp_griffin_lim_ragged(one, one, None, 2, 90, n_fft, hop, win, 1, one, one, one, one, None) == -1   # no host copy
