"""Op-level hunt for the run-to-run differences of the bf16 storage engine's backward in multi-stream mode (profiles/r06_bf16_repro*.txt): each
pair-tensor kernel of the content encoder's T = 128 levels is launched N times on one stream -- alone, and beside a second stream that keeps
the chip busy with another of this library's kernels -- and every result is compared bit for bit with the first.
usage: python scripts/pairs_race_probe.py [B] [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
C = 128


def pairs(B, C, T, scale=1.0):
    return (scale * torch.randn(B, C, T, device=dev)).to(torch.bfloat16).view(torch.int32).view(B, C // 2, T)


def pack(w, dgrad):
    Cout, Cin, KS = w.shape
    n = lib.avc_packed_weight_floats(Cout, Cin, KS, dgrad)
    dst = torch.zeros(n, device=dev)
    arr = (ctypes.c_void_p * 1)(w.data_ptr())
    assert lib.avc_pack_weight(arr, 1, Cout, Cout, Cin, KS, dgrad, P(dst), None) == 0
    return dst


torch.manual_seed(0)
lib.avc_set_tuning(b"op_compute_dtype", 3)
w = torch.randn(C, C, 5, device=dev) / (C * 5) ** 0.5
wpf, wpd = pack(w, 0), pack(w, 1)
bias = torch.randn(C, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
stA, stB = ctypes.c_void_p(sA.cuda_stream), ctypes.c_void_p(sB.cuda_stream)

# background load: a big fp32-storage conv of this library on the second stream
lib.avc_set_tuning(b"op_compute_dtype", 0)
wbg = torch.randn(C, C, 5, device=dev) / (C * 5) ** 0.5
wbgp = pack(wbg, 0)
lib.avc_set_tuning(b"op_compute_dtype", 3)
xbg = torch.randn(64, C, 128, device=dev)
obg = torch.zeros_like(xbg)


MODE = sys.argv[3] if len(sys.argv) > 3 else "fp32conv"
dyb_, maskb_ = pairs(B, C, 128, 1e-3), pairs(B, C, 128)
dxb_, dxb2_ = torch.zeros_like(dyb_), torch.zeros_like(dyb_)
resb_ = pairs(B, C, 128, 1e-3)


def bg():
    if MODE == "fp32conv":
        lib.avc_set_tuning(b"op_compute_dtype", 0)
        lib.avc_conv1d_fwd(P(xbg), xbg.stride(0), xbg.stride(1), 1, 64, C, 128, P(wbgp), None, C, 5, 1, 1, P(obg), obg.stride(0), obg.stride(1), 1, 1, None, 0, 0, 0, 0, 0, None, 0, stB)
        lib.avc_set_tuning(b"op_compute_dtype", 3)
    else:   # what the speaker branch of the backward pass runs beside the content branch: pair input gradients with a masked second output
        lib.avc_conv1d_dgrad(P(dyb_), dyb_.stride(0), dyb_.stride(1), 1, 1, B, C, 128, P(wpd), C, 5, 1, 128, None, dxb_.stride(0), dxb_.stride(1), 1, None, 0, 0, 0, 0, 0, P(dxb2_), P(maskb_), 0, stB)
        lib.avc_conv1d_dgrad(P(dxb2_), dyb_.stride(0), dyb_.stride(1), 1, 1, B, C, 128, P(wpd), C, 5, 1, 128, P(dxb_), dxb_.stride(0), dxb_.stride(1), 1, P(resb_), 1, resb_.stride(0), resb_.stride(1), 1, 128, P(dxb2_), P(maskb_), 0, stB)


def trial(name, launch, outs):
    for load in (False, True):
        for o in outs:
            o.zero_()
        assert launch() == 0
        torch.cuda.synchronize()
        ref = [o.clone() for o in outs]
        bad = torch.zeros((), device=dev, dtype=torch.int32)
        nb = 0
        for i in range(N):
            if load:
                bg(); bg()
            for o in outs:
                with torch.cuda.stream(sA):
                    o.zero_()
            assert launch() == 0
            with torch.cuda.stream(sA):
                for o, r in zip(outs, ref):
                    bad += (o != r).any().int()
        torch.cuda.synchronize()
        print(f"{name:64s} {'beside a busy stream' if load else 'alone':22s}: {int(bad.item())} of {N} launches differ from the first", flush=True)
        if int(bad.item()) and len(sys.argv) > 4:   # look at ONE differing launch: synchronise after each until one differs
            for i in range(4 * N):
                bg(); bg()
                with torch.cuda.stream(sA):
                    for o in outs:
                        o.zero_()
                assert launch() == 0
                sA.synchronize()
                d = [(o != r) for o, r in zip(outs, ref)]
                if any(bool(x.any()) for x in d):
                    for k, (x, o, r) in enumerate(zip(d, outs, ref)):
                        idx = x.flatten().nonzero().flatten()
                        if len(idx):
                            fo, fr = o.flatten(), r.flatten()
                            print(f"    launch {i}: output {k}: {len(idx)} elements differ; indices {idx[:12].tolist()} ... {idx[-3:].tolist()}; got {[hex(int(v) & 0xffffffff) for v in fo[idx[:6]].view(torch.int32).tolist()]} expected {[hex(int(v) & 0xffffffff) for v in fr[idx[:6]].view(torch.int32).tolist()]}")
                    torch.cuda.synchronize()
                    # did anything but the output change?  (the launch's own inputs are checked by the caller)
                    break


for T in (128, 64):
    x, dy = pairs(B, C, T), pairs(B, C, T, 1e-3)
    out, dx = torch.zeros_like(x), torch.zeros_like(x)
    res = pairs(B, C, T, 1e-3)
    trial(f"conv fwd k5 128->128 T={T}", lambda: lib.avc_conv1d_fwd(P(x), x.stride(0), x.stride(1), 1, B, C, T, P(wpf), P(bias), C, 5, 1, 0, P(out), out.stride(0), out.stride(1), 1, 1, None, 0, 0, 0, 0, 0, None, 0, stA), [out])
    trial(f"conv dgrad k5 stride 1 (reflect adjoint) T={T}", lambda: lib.avc_conv1d_dgrad(P(dy), dy.stride(0), dy.stride(1), 1, 1, B, C, T, P(wpd), C, 5, 1, T, P(dx), dx.stride(0), dx.stride(1), 1, None, 0, 0, 0, 0, 0, None, None, 0, stA), [dx])
    trial(f"conv dgrad k5 stride 1 + identity residual T={T}", lambda: lib.avc_conv1d_dgrad(P(dy), dy.stride(0), dy.stride(1), 1, 1, B, C, T, P(wpd), C, 5, 1, T, P(dx), dx.stride(0), dx.stride(1), 1, P(res), 1, res.stride(0), res.stride(1), 1, T, None, None, 0, stA), [dx])
    dy2 = pairs(B, C, T // 2, 1e-3)
    trial(f"conv dgrad k5 stride 2 (column parity per wave) Tin={T}", lambda: lib.avc_conv1d_dgrad(P(dy2), dy2.stride(0), dy2.stride(1), 1, 1, B, C, T // 2, P(wpd), C, 5, 2, T, P(dx), dx.stride(0), dx.stride(1), 1, None, 0, 0, 0, 0, 0, None, None, 0, stA), [dx])
    y = pairs(B, C, T)
    g = pairs(B, C, T, 1e-3)
    o2 = torch.zeros_like(y)
    mean, rstd = torch.zeros(B * C, device=dev), torch.zeros(B * C, device=dev)
    trial(f"instnorm fwd pairs T={T}", lambda: lib.avc_instnorm_fwd_pairs(P(y), B, C, T, None, 0, 0, 1, None, 0, 0, 0, P(o2), P(mean), P(rstd), stA), [o2, mean, rstd])
    keep = [t.clone() for t in (g, y, mean, rstd)]
    trial(f"instnorm bwd pairs T={T}", lambda: lib.avc_instnorm_bwd_pairs(P(g), P(y), P(mean), P(rstd), B, C, T, None, 0, 0, 1, 0, P(o2), None, 0, 0, stA), [o2])
    print("    inputs of the instnorm backward unchanged afterwards:", [bool(torch.equal(a_, b_)) for a_, b_ in zip(keep, (g, y, mean, rstd))],
          " addresses g/y/mean/rstd/out:", [hex(t.data_ptr()) for t in (g, y, mean, rstd, o2)], " background out/out2/mask/res/dy:", [hex(t.data_ptr()) for t in (dxb_, dxb2_, maskb_, resb_, dyb_)])
    ws = torch.zeros(lib.avc_conv1d_wgrad_ws_floats(B, C, C, T, 5), device=dev)
    dW, db = torch.zeros(C, C, 5, device=dev), torch.zeros(C, device=dev)
    trial(f"wgrad k5 128->128 T={T}", lambda: lib.avc_conv1d_wgrad(P(x), x.stride(0), x.stride(1), 1, P(dy), dy.stride(0), dy.stride(1), 1, 1, B, C, C, T, T, 5, 1, P(dW), P(db), P(ws), stA), [dW, db])
