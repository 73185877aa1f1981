"""Ablation of the conv / weight-gradient kernels on bf16 PAIR tensors (GPU only; op_compute_dtype 3): where a launch spends its time
once the matrix products take 1/8 of the fp32 time.  conv_ablation bits: 1 no LDS-DMA after the first chunks, 2 no MFMA / LDS reads,
4 no barrier, 8 no store.  Results are wrong by construction; timing only."""
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
    return torch.randn(B, C, T, device=dev).to(torch.bfloat16).view(torch.int32).view(B, C // 2, T)   # (any finite bf16 pairs)


def pack(w, dgrad):
    Cout, Cin, KS = w.shape
    n = lib.avc_packed_weight_floats(Cout, Cin, KS, dgrad)
    dst = torch.zeros(n, device=dev)
    arr = (ctypes.c_void_p * 1)(w.data_ptr())
    assert lib.avc_pack_weight(arr, 1, Cout, Cout, Cin, KS, dgrad, P(dst), None) == 0
    return dst


def run(B, Cin, Cout, T, KS, tile, ck, mode="f"):
    lib.avc_set_tuning(b"op_compute_dtype", 3)
    lib.avc_set_tuning(b"conv_ck5", ck)
    x, dy = pairs(B, Cin, T), pairs(B, Cout, T)
    w = torch.randn(Cout, Cin, KS, device=dev) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, device=dev)
    out, dx = torch.zeros_like(dy), torch.zeros_like(x)
    wp, wpd = pack(w, 0), pack(w, 1)
    res = []
    for dbg, name in ((0, "full"), (1, "noDMA"), (2, "noMFMA"), (7, "empty"), (15, "empty,noEpi")):
        lib.avc_set_tuning(b"conv_ablation", dbg)
        if mode == "f":
            f = lambda: lib.avc_conv1d_fwd(P(x), x.stride(0), x.stride(1), 1, B, Cin, T, P(wp), P(b), Cout, KS, 1, 1, P(out), out.stride(0), out.stride(1), 1, 1,
                                           None, 0, 0, 0, 0, 0, None, tile, None)
        else:
            f = lambda: lib.avc_conv1d_dgrad(P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cout, T, P(wpd), Cin, KS, 1, T, P(dx), dx.stride(0), dx.stride(1), 1,
                                             None, 0, 0, 0, 0, 0, None, None, tile, None)
        rc = f()
        if rc != 0:
            res.append(f"{name}: rc={rc}")
            break
        res.append(f"{name}: {timeit(f):6.1f}us")
    lib.avc_set_tuning(b"conv_ablation", 0)
    lib.avc_set_tuning(b"conv_ck5", 8)
    lib.avc_set_tuning(b"op_compute_dtype", 0)
    print(f"{mode} B={B} {Cin}->{Cout} T={T} k={KS} t{tile} ck{ck}: " + " | ".join(res), flush=True)


def run_wgrad(B, Cin, Cout, T, KS):
    lib.avc_set_tuning(b"op_compute_dtype", 3)
    x, dy = pairs(B, Cin, T), pairs(B, Cout, T)
    ws = torch.zeros(lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, T, KS), device=dev)
    dW, db = torch.zeros(Cout, Cin, KS, device=dev), torch.zeros(Cout, device=dev)
    res = []
    for dbg, name in ((0, "full"), (1, "noDMA"), (2, "noMFMA"), (3, "neither"), (7, "neither,noBar"), (15, "empty")):
        lib.avc_set_tuning(b"wgrad_ablation", dbg)
        f = lambda: lib.avc_conv1d_wgrad(P(x), x.stride(0), x.stride(1), 1, P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cin, Cout, T, T, KS, 1, P(dW), P(db), P(ws), None)
        assert f() == 0
        res.append(f"{name}: {timeit(f):6.1f}us")
    lib.avc_set_tuning(b"wgrad_ablation", 0)
    lib.avc_set_tuning(b"op_compute_dtype", 0)
    print(f"wgrad(+reduce) B={B} {Cin}->{Cout} T={T} k={KS}: " + " | ".join(res), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "in":
    sys.argv = sys.argv[:1]
    import bench
    for nv in (1, 2, 4):
        lib.avc_set_tuning(b"in_pairs_nv", nv)
        for T in (128, 64, 32, 1024):
            B = 256 if T <= 128 else 64
            r = bench.instnorm_dominant_shape(B, 128, T, pairs=True)
            print(f"IN pairs nv={nv} [{B},128,{T}] fwd {r['fwd']['avg_launch_us']:.1f}us {r['fwd']['gbs']:.0f} GB/s bwd {r['bwd']['avg_launch_us']:.1f}us {r['bwd']['gbs']:.0f} GB/s", flush=True)
    lib.avc_set_tuning(b"in_pairs_nv", 1)
    sys.exit(0)

if __name__ == "__main__":
    B = 256
    for ck in (8, 16, 32):
        run(B, 128, 128, 128, 5, 11, ck)
    run(B, 128, 128, 128, 5, 21, 16)
    run(B, 128, 128, 128, 5, 11, 16, "d")
    run(B, 128, 128, 128, 5, 11, 8, "d")
    run(B, 128, 128, 32, 5, 11, 16)
    run(B, 128, 128, 16, 5, 11, 16)
    run(B, 1104, 128, 128, 1, 21, 8)
    run_wgrad(B, 128, 128, 128, 5)
    run_wgrad(B, 128, 128, 32, 5)
    run_wgrad(B, 1104, 128, 128, 1)
    # InstanceNorm pair rows over the shapes of a step
    sys.argv = sys.argv[:1]
    import bench
    for T in (128, 64, 32, 16):
        r = bench.instnorm_dominant_shape(B, 128, T, pairs=True)
        r32 = bench.instnorm_dominant_shape(B, 128, T, pairs=False)
        print(f"IN [256,128,{T}] pairs fwd {r['fwd']['avg_launch_us']:.1f}us {r['fwd']['gbs']:.0f} GB/s bwd {r['bwd']['avg_launch_us']:.1f}us {r['bwd']['gbs']:.0f} GB/s | "
              f"fp32 fwd {r32['fwd']['avg_launch_us']:.1f}us {r32['fwd']['gbs']:.0f} GB/s bwd {r32['bwd']['avg_launch_us']:.1f}us {r32['bwd']['gbs']:.0f} GB/s", flush=True)
