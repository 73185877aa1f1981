"""Ablation of the conv kernel's main loop (GPU only): avc_set_tuning("conv_ablation") conv bit0 = no LDS-DMA after the first
chunks, bit1 = no MFMA/LDS reads, bit2 = no barrier.  Results are wrong by construction; timing only."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
from conv_micro import pack, timeit

def run(B, Cin, Cout, T, KS, tiles, mode="f"):
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, KS, device=dev) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, device=dev)
    out = torch.zeros(B, Cout, T, device=dev)
    dy = torch.randn(B, Cout, T, device=dev)
    dx = torch.zeros(B, Cin, T, device=dev)
    wp, wpd = pack(w, 0), pack(w, 1)
    flops = 2.0 * Cout * Cin * KS * B * T
    for tile in tiles:
        res = []
        for dbg, name in ((0, "full"), (1, "noDMA"), (5, "noDMA,noBar"), (2, "noMFMA"), (7, "empty"), (15, "empty,noEpi")):
            lib.avc_set_tuning(b"conv_ablation", dbg)
            if mode == "f":
                f = lambda: lib.avc_conv1d_fwd(P(x), x.stride(0), x.stride(1), 1, B, Cin, T, P(wp), P(b), Cout, KS, 1, 1, P(out),
                                               out.stride(0), out.stride(1), 1, 1, None, 0, 0, 0, 0, 0, None, tile, None)
            else:
                f = lambda: lib.avc_conv1d_dgrad(P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cout, T, P(wpd), Cin, KS, 1, T, P(dx),
                                                 dx.stride(0), dx.stride(1), 1, None, 0, 0, 0, 0, 0, None, None, tile, None)
            assert f() == 0
            us = timeit(f)
            res.append(f"{name}: {us:6.1f}us")
        (lib.avc_set_tuning(b"conv_ablation", 0), lib.avc_set_tuning(b"wgrad_ablation", 0))
        print(f"{mode} B={B} {Cin}->{Cout} T={T} k={KS} t{tile} (ideal {flops/157.3e6:5.1f}us): " + " | ".join(res), flush=True)

def run_x3(B, Cin, Cout, T):
    from conv_micro import pack_x3
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, 5, device=dev) / (Cin * 5) ** 0.5
    b = torch.randn(Cout, device=dev)
    out = torch.zeros(B, Cout, T, device=dev)
    wp = pack_x3(w, 0)
    res = []
    for dbg, name in ((0, "full"), (1, "noDMA"), (2, "noMFMA/split"), (3, "neither")):
        lib.avc_set_tuning(b"conv_ablation", dbg)
        f = lambda: lib.avc_conv1d_fwd(P(x), x.stride(0), x.stride(1), 1, B, Cin, T, P(wp), P(b), Cout, 5, 1, 1, P(out), out.stride(0), out.stride(1), 1, 1,
                                       None, 0, 0, 0, 0, 0, None, 97, None)
        assert f() == 0
        res.append(f"{name}: {timeit(f):6.1f}us")
    (lib.avc_set_tuning(b"conv_ablation", 0), lib.avc_set_tuning(b"wgrad_ablation", 0))
    print(f"x3 fwd B={B} {Cin}->{Cout} T={T}: " + " | ".join(res), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "x3":
    run_x3(256, 128, 128, 128)
    run_x3(1024, 128, 128, 128)
    run_x3(256, 128, 128, 64)
    sys.exit(0)

if __name__ == "__main__":
    B = 256
    run(B, 128, 128, 128, 5, (11,))
    run(B, 128, 128, 128, 5, (11,), "d")
    run(B, 128, 128, 32, 5, (11,))
    run(B, 1104, 128, 128, 1, (21,))
    run(B, 1024, 128, 128, 1, (21,), "d")   # in_conv dgrad shape: K = 128, M = 1024
    run(1, 128, 128, 64, 1, (11,))
