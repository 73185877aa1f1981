"""Ablation of the wgrad kernel (GPU only): avc_set_tuning("wgrad_ablation") wgrad bit0 no DMA after first chunk, bit1 no MFMA,
bit2 no barrier, bit3 no slab store.  Timing only (results wrong by construction)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
from conv_micro import timeit

def run(B, Cin, Cout, T, KS):
    x = torch.randn(B, Cin, T, device=dev)
    dy = torch.randn(B, Cout, T, device=dev)
    ws = torch.zeros(lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, T, KS), device=dev)
    dW = torch.zeros(Cout, Cin, KS, device=dev); db = torch.zeros(Cout, device=dev)
    flops = 2.0 * Cout * Cin * KS * B * T
    res = []
    for dbg, name in ((0, "full"), (1, "noDMA"), (5, "noDMA,noBar"), (2, "noMFMA"), (8, "noStore"), (15, "empty"), (7, "store only")):
        lib.avc_set_tuning(b"wgrad_ablation", dbg)
        f = lambda: lib.avc_conv1d_wgrad(P(x), x.stride(0), x.stride(1), 1, P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cin, Cout, T, T,
                                         KS, 1, P(dW), P(db), P(ws), None)
        assert f() == 0
        res.append(f"{name}: {timeit(f):6.1f}us")
    (lib.avc_set_tuning(b"conv_ablation", 0), lib.avc_set_tuning(b"wgrad_ablation", 0))
    print(f"wgrad+reduce B={B} {Cin}->{Cout} T={T} k={KS} (ideal {flops/157.3e6:5.1f}us): " + " | ".join(res), flush=True)

if __name__ == "__main__":
    B = 256
    run(B, 128, 128, 128, 5)
    run(B, 128, 128, 64, 5)
    run(B, 128, 128, 32, 5)
    run(B, 128, 128, 16, 5)
    run(B, 1104, 128, 128, 1)
    run(B, 80, 128, 128, 8)
    run(B, 80, 128, 128, 3)
    run(1, 128, 128, 256, 1)
