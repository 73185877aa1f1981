"""Micro-benchmark of the conv kernels at model shapes through the op-level C ABI (GPU only)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())

def pack(w, dgrad):
    Cout, Cin, KS = w.shape
    n = lib.avc_packed_weight_floats(Cout, Cin, KS, dgrad)
    dst = torch.zeros(n, device=dev)
    arr = (ctypes.c_void_p * 1)(w.data_ptr())
    assert lib.avc_pack_weight(arr, 1, Cout, Cout, Cin, KS, dgrad, P(dst), None) == 0
    return dst

def pack_x3(w, dgrad):
    Cout, Cin, KS = w.shape
    dst = torch.zeros(lib.avc_packed_weight_floats_x3(Cout, Cin, KS, dgrad), device=dev)
    assert lib.avc_pack_weight_x3(P(w), Cout, Cin, KS, dgrad, P(dst), None) == 0
    return dst

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

def run(B, Cin, Cout, T, KS, stride, tiles=(21, 11), which="fdw"):
    x = torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, KS, device=dev) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, device=dev)
    padL, padR = KS // 2, (KS // 2 - 1 if KS % 2 == 0 else KS // 2)
    To = (T + padL + padR - KS) // stride + 1
    out = torch.zeros(B, Cout, To, device=dev)
    dy = torch.randn(B, Cout, To, device=dev)
    dx = torch.zeros(B, Cin, T, device=dev)
    wp, wpd = pack(w, 0), pack(w, 1)
    x3_ok = ((KS == 5 and Cin % 16 == 0 and Cout % 16 == 0) or (KS == 1 and Cin >= 32 and Cout >= 32)) and 97 in tiles
    wx3, wx3d = (pack_x3(w, 0), pack_x3(w, 1)) if x3_ok else (None, None)
    flops = 2.0 * Cout * Cin * KS * B * To
    res = []
    for tile in tiles:
        if "f" in which:
            f = lambda: lib.avc_conv1d_fwd(P(x), x.stride(0), x.stride(1), 1, B, Cin, T, P(wx3 if tile == 97 else wp), P(b), Cout, KS, stride, 1, P(out),
                                           out.stride(0), out.stride(1), 1, 1, None, 0, 0, 0, 0, 0, None, tile, None)
            assert f() == 0
            us = timeit(f); res.append(f"fwd t{tile}: {us:7.1f}us {flops/us/1e6:6.1f}TF")
        if "d" in which:
            f = lambda: lib.avc_conv1d_dgrad(P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cout, To, P(wx3d if tile == 97 else wpd), Cin, KS, stride, T, P(dx),
                                             dx.stride(0), dx.stride(1), 1, None, 0, 0, 0, 0, 0, None, None, tile, None)
            assert f() == 0
            us = timeit(f); res.append(f"dgr t{tile}: {us:7.1f}us {flops/us/1e6:6.1f}TF")
    if "w" in which:
        ws = torch.zeros(lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, To, KS), device=dev)
        dW = torch.zeros(Cout, Cin, KS, device=dev); db = torch.zeros(Cout, device=dev)
        f = lambda: lib.avc_conv1d_wgrad(P(x), x.stride(0), x.stride(1), 1, P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cin, Cout, T, To,
                                         KS, stride, P(dW), P(db), P(ws), None)
        assert f() == 0
        us = timeit(f); res.append(f"wgrad(+reduce): {us:7.1f}us {flops/us/1e6:6.1f}TF")
    print(f"B={B} {Cin}->{Cout} T={T} k={KS} s={stride}: " + " | ".join(res), flush=True)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "x3":
    for B in (256, 1024):
        for T in (128, 64, 32, 16):
            run(B, 128, 128, T, 5, 1, tiles=(0, 97), which="fd")
        run(B, 128, 128, 128, 5, 2, tiles=(0, 97), which="fd")
        run(B, 128, 256, 64, 5, 1, tiles=(0, 97), which="f")
    run(64, 128, 128, 1024, 5, 1, tiles=(0, 97), which="fd")
    run(256, 1104, 128, 128, 1, 1, tiles=(0, 97), which="fd")
    run(256, 128, 1024, 128, 1, 1, tiles=(0, 97), which="f")
    run(256, 128, 128, 128, 1, 1, tiles=(0, 97), which="fd")
    run(1, 1200, 2050, 401, 1, 1, tiles=(0, 97), which="f")
    run(1, 2050, 1200, 401, 1, 1, tiles=(0, 97), which="f")
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "s2":
    for par in (0, 1):
        lib.avc_set_tuning(b"dgrad_par", par)
        print("dgrad_par", par)
        for T in (128, 64, 32):
            run(256, 128, 128, T, 5, 2, tiles=(0,), which="d")
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 0 and sys.argv[0] != "x":
    B = 256
    run(B, 128, 128, 128, 5, 1, tiles=(21, 11, 12))
    run(B, 128, 128, 64, 5, 1, tiles=(21, 11, 12))
    run(1024, 128, 128, 128, 5, 1, tiles=(21, 11, 12), which="fd")
    run(1024, 128, 128, 64, 5, 1, tiles=(21, 11, 12), which="fd")
    run(B, 128, 128, 32, 5, 1, tiles=(11,))
    run(B, 128, 128, 16, 5, 1, tiles=(11,))
    run(B, 128, 128, 128, 5, 2, tiles=(21, 11))
    run(B, 1104, 128, 128, 1, 1, tiles=(21, 11))
    run(B, 80, 128, 128, 8, 1, tiles=(21, 11), which="fw")
    run(B, 128, 1024, 128, 1, 1, tiles=(21, 11), which="f")
    run(1, 128, 128, 256, 1, 1, tiles=(11,))
