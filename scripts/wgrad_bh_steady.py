"""Steady-state rate of the bf16 pair-storage weight gradient (GPU only): ONE k = 5 128 -> 128 layer at a batch large enough that the fixed
cost of a launch (prologue, partial stores, reduce) is noise and x / dy (2 x B x 128 x T x 2 bytes) exceed the caches -- per-chunk time of a
workgroup in each ablation.  Run once per build (AVC_HIP_LIB) to compare producer ring depths.
usage: python scripts/wgrad_bh_steady.py [B ...]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from adaptive_voice_conversion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
from conv_micro import timeit


def pairs(B, C, T):
    return torch.randn(B, C, T, device=dev).to(torch.bfloat16).view(torch.int32).view(B, C // 2, T)


def run(B, Cin, Cout, T, KS):
    lib.avc_set_tuning(b"op_compute_dtype", 3)
    x, dy = pairs(B, Cin, T), pairs(B, Cout, T)
    ws = torch.zeros(lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, T, KS), device=dev)
    dW, db = torch.zeros(Cout, Cin, KS, device=dev), torch.zeros(Cout, device=dev)
    res = []
    tiles = 4 if Cin % 64 == 0 else 3 * (Cout // 128)   # 64 x 64 tiles, or 128 co x 32 ci (Cin = 80: three ci tiles)
    chunks_per_wg = tiles * (B * T // 32) / 256.0
    for dbg, name in ((0, "full"), (2, "noMFMA"), (1, "noDMA"), (3, "neither"), (15, "empty")):
        lib.avc_set_tuning(b"wgrad_ablation", dbg)
        f = lambda: lib.avc_conv1d_wgrad(P(x), x.stride(0), x.stride(1), 1, P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cin, Cout, T, T, KS, 1, P(dW), P(db), P(ws), None)
        assert f() == 0
        us = timeit(f)
        res.append(f"{name}: {us:7.1f}us ({1e3 * us / chunks_per_wg:5.0f} ns/chunk)")
    lib.avc_set_tuning(b"wgrad_ablation", 0)
    lib.avc_set_tuning(b"op_compute_dtype", 0)
    flops = 2.0 * Cout * Cin * KS * B * T
    print(f"wgrad bf16s B={B} {Cin}->{Cout} T={T} k={KS} ({2 * B * 128 * T * 2 / 1e6:.0f} MB of x + dy): " + " | ".join(res), flush=True)


if __name__ == "__main__":
    print("lib:", os.environ.get("AVC_HIP_LIB", "default"))
    for B in ([int(v) for v in sys.argv[1:]] or [256, 2048, 8192]):
        run(B, 128, 128, 128, 5)
    run(2048, 128, 128, 32, 5)
    run(2048, 128, 128, 16, 5)
    for k in (8, 4, 1):
        run(2048, 80, 128, 128, k)     # a conv-bank member (the run-time-taps instance for k != 1, 5)
