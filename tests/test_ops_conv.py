"""Implicit-GEMM conv kernels (forward / dgrad / wgrad) vs torch restatements of
model.py:21-32 (pad_layer) and its autograd.

kind='emu' runs the same .hip sources on the CPU lane-level simulator (kernel
logic, runs anywhere); kind='gpu' calls the gfx950 library through the C ABI.
Tolerance: fp32, rtol 1e-5/1e-4 (MFMA == fmaf chain, only the summation order
differs from MKL-DNN)."""
import ctypes

import pytest
import torch

from oracle import avc_oracle as O
from tests.emu_util import KINDS, P, backend

GPU = pytest.mark.gpu


def pack(lib, dev, ws, dgrad):
    Cout, Cin, KS = ws[0].shape[0] * len(ws), ws[0].shape[1], (ws[0].shape[2] if ws[0].dim() == 3 else 1)
    n = lib.avc_packed_weight_floats(Cout, Cin, KS, dgrad)
    dst = torch.full((n,), float("nan"), device=dev)
    arr = (ctypes.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    assert lib.avc_pack_weight(arr, len(ws), ws[0].shape[0], Cout, Cin, KS, dgrad, P(dst), None) == 0
    assert torch.isfinite(dst).all()
    return dst


def conv_fwd(lib, dev, x, w, b, stride=1, act=0, tile=0, res=None, res_mode=0, ops=1):
    B, Cin, Tin = x.shape
    Cout, _, KS = w.shape
    wp = pack(lib, dev, [w], 0)
    padL, padR = KS // 2, (KS // 2 - 1 if KS % 2 == 0 else KS // 2)
    Tout = (Tin + padL + padR - KS) // stride + 1
    out = torch.full((B, Cout // ops, Tout * ops), float("nan"), device=dev)
    out2 = torch.full_like(out, float("nan")) if res is not None else None
    rb = rc = rt = Tres = 0
    if res is not None:
        rb, rc, rt, Tres = res.stride(0), res.stride(1), res.stride(2), res.shape[2]
    rcode = lib.avc_conv1d_fwd(P(x), x.stride(0), x.stride(1), x.stride(2), B, Cin, Tin, P(wp), P(b), Cout, KS, stride,
                               act, P(out), out.stride(0), out.stride(1), out.stride(2), ops, P(res), res_mode, rb, rc,
                               rt, Tres, P(out2), tile, None)
    assert rcode == 0, rcode
    return out, out2


FWD = [
    # B, Cin, Cout, T, KS, stride, tile
    (2, 16, 32, 32, 5, 1, 11),
    (3, 24, 40, 20, 5, 2, 11),
    (1, 16, 32, 70, 5, 1, 11),    # Tout > BN, partial last tile
    (2, 8, 32, 19, 8, 1, 11),     # even kernel (bank), odd T
    (2, 8, 32, 17, 2, 1, 11),
    (1, 40, 32, 33, 1, 1, 11),    # 1x1
    (2, 16, 130, 64, 3, 1, 21),   # 128x64 tile, 2 M tiles
    (1, 16, 128, 130, 5, 1, 21),  # 128-row tile, three column tiles, ragged last one
    (1, 16, 64, 200, 5, 1, 12),   # 64 x 128 tile (two column fragments per wave), ragged second tile
    (3, 24, 40, 64, 5, 1, 12),    # ... two whole samples per tile, odd batch
    (2, 20, 32, 32, 5, 1, 11),    # reduction channels not a multiple of 8: the last 8-channel unit straddles Cin
    (2, 44, 32, 24, 3, 1, 11),    # ... with 16-channel chunks: a whole padded unit + a straddling one
    (5, 16, 32, 3, 5, 1, 11),     # bottleneck-sized rows
    (2, 16, 32, 21, 5, 2, 11),    # stride 2, odd T
    (2, 40, 32, 32, 5, 1, 11),    # 5 K-chunks: two split-K wave groups with unequal chunk counts
    (3, 64, 64, 16, 5, 1, 11),    # 8 K-chunks, several short samples per tile, split-K
    (5, 128, 40, 16, 5, 1, 0),    # short rows, launcher's own choice of tile / chunk depth / split-K groups
    (9, 128, 128, 32, 5, 1, 0),
    pytest.param(8, 128, 128, 128, 5, 1, 0, marks=GPU),
    pytest.param(8, 128, 128, 128, 5, 1, 12, marks=GPU),
    pytest.param(8, 128, 128, 128, 5, 2, 0, marks=GPU),
    pytest.param(8, 128, 256, 64, 5, 1, 0, marks=GPU),
    pytest.param(64, 128, 128, 16, 5, 1, 0, marks=GPU),
    pytest.param(4, 1104, 128, 128, 1, 1, 0, marks=GPU),
    pytest.param(4, 80, 128, 128, 8, 1, 21, marks=GPU),
    pytest.param(2, 128, 128, 1024, 5, 1, 21, marks=GPU),
    pytest.param(2, 512, 128, 128, 7, 1, 21, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,tile", FWD)
def test_conv_fwd_matches_pad_conv(kind, B, Cin, Cout, T, KS, stride, tile):
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = torch.relu(O.pad_conv(x, w, b, stride))
    out, _ = conv_fwd(lib, dev, x.to(dev), w.to(dev), b.to(dev), stride, act=1, tile=tile)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("kind", KINDS)
def test_conv_fwd_transposed_input_view_and_residual_pool(kind):
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(7)
    B, Cin, Cout, T = 2, 16, 32, 22
    xt = torch.randn(B, T, Cin, generator=g)
    x = xt.transpose(1, 2)  # the [B,M,T] view of data_utils.py:14-16 (strides (T*M, 1, M))
    w = torch.randn(Cout, Cin, 5, generator=g) / 9
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    y = torch.relu(O.pad_conv(x, w, b, 2))
    out, out2 = conv_fwd(lib, dev, xt.to(dev).transpose(1, 2), w.to(dev), b.to(dev), 2, act=1, tile=11, res=res.to(dev), res_mode=2)
    torch.testing.assert_close(out.cpu(), y, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out2.cpu(), y + O.avg_pool_ceil(res, 2), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind", KINDS)
def test_conv_fwd_pixel_shuffle_store(kind):
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 16, 12, generator=g)
    w = torch.randn(64, 16, 5, generator=g) / 9
    b = torch.randn(64, generator=g)
    ref = O.pixel_shuffle_1d(O.pad_conv(x, w, b), 2)
    out, _ = conv_fwd(lib, dev, x.to(dev), w.to(dev), b.to(dev), 1, act=0, tile=11, ops=2)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)


DG = [
    (2, 16, 32, 32, 5, 1, 11),
    (3, 24, 40, 20, 5, 2, 11),
    (2, 16, 32, 21, 5, 2, 11),
    (1, 16, 32, 70, 5, 1, 11),
    (1, 16, 32, 130, 5, 1, 11),   # right mirror spans the last two tiles
    (1, 40, 32, 33, 1, 1, 11),
    (5, 16, 32, 3, 5, 1, 11),
    (1, 16, 128, 130, 5, 1, 21),
    (2, 16, 32, 66, 5, 1, 21),
    (1, 16, 32, 200, 5, 1, 12),   # 64 x 128 tile: mirror windows in both column fragments
    (3, 16, 32, 64, 5, 1, 12),
    (2, 20, 32, 32, 5, 1, 11),    # reduction channels (Cout here) not a multiple of 8
    (5, 40, 128, 16, 5, 1, 0),    # mirror windows of 4 samples per tile
    (3, 16, 32, 32, 5, 2, 11),    # stride 2 with one column parity per wave (even taps / odd taps): two samples per tile
    (5, 16, 32, 16, 5, 2, 11),    # ... four samples per tile, ragged last tile
    (1, 16, 32, 128, 5, 2, 11),   # ... two tiles per sample
    (1, 16, 32, 101, 5, 2, 11),   # ... odd length, partial last tile
    (2, 64, 32, 32, 5, 2, 11),    # ... with the two split-K wave groups
    (3, 128, 128, 32, 5, 1, 0),
    pytest.param(8, 128, 128, 128, 5, 1, 0, marks=GPU),
    pytest.param(8, 128, 128, 128, 5, 2, 0, marks=GPU),
    pytest.param(64, 128, 128, 16, 5, 1, 0, marks=GPU),
    pytest.param(2, 128, 128, 1024, 5, 1, 0, marks=GPU),
    pytest.param(4, 128, 80, 128, 1, 1, 0, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,tile", DG)
def test_conv_dgrad_matches_autograd(kind, B, Cin, Cout, T, KS, stride, tile):
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 77 + T)
    x = torch.randn(B, Cin, T, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    y = O.pad_conv(x, w, None, stride)
    dy = torch.randn(y.shape, generator=g)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    wpd = pack(lib, dev, [w.to(dev)], 1)
    dx = torch.full((B, Cin, T), float("nan"), device=dev)
    dyd = dy.to(dev)
    rc = lib.avc_conv1d_dgrad(P(dyd), dyd.stride(0), dyd.stride(1), dyd.stride(2), 1, B, Cout, dy.shape[2], P(wpd), Cin,
                              KS, stride, T, P(dx), dx.stride(0), dx.stride(1), dx.stride(2), None, 0, 0, 0, 0, 0, None,
                              None, tile, None)
    assert rc == 0
    torch.testing.assert_close(dx.cpu(), dx_ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("T,ck", [(16, 16), (32, 16), (32, 32)])
def test_conv_with_deeper_chunks(kind, T, ck):
    """The plan gives layers that cannot fill the chip 16- or 32-channel K-chunks (conv_ck5 at the op level): same function."""
    lib, dev = backend(kind)
    assert lib.avc_set_tuning(b"conv_ck5", ck) == 0
    try:
        g = torch.Generator().manual_seed(T)
        B, C = 5, 128
        x = torch.randn(B, C, T, generator=g, requires_grad=True)
        w = torch.randn(C, C, 5, generator=g) / (C * 5) ** 0.5
        b = torch.randn(C, generator=g)
        y = O.pad_conv(x, w, b, 1)
        dy = torch.randn(y.shape, generator=g)
        (dx_ref,) = torch.autograd.grad(y, x, dy)
        out, _ = conv_fwd(lib, dev, x.detach().to(dev), w.to(dev), b.to(dev), 1, act=0, tile=11)
        torch.testing.assert_close(out.cpu(), y.detach(), rtol=1e-5, atol=2e-5)
        wpd = pack(lib, dev, [w.to(dev)], 1)
        dx = torch.full((B, C, T), float("nan"), device=dev)
        dyd = dy.to(dev)
        rc = lib.avc_conv1d_dgrad(P(dyd), dyd.stride(0), dyd.stride(1), dyd.stride(2), 1, B, C, T, P(wpd), C, 5, 1, T, P(dx),
                                  dx.stride(0), dx.stride(1), dx.stride(2), None, 0, 0, 0, 0, 0, None, None, 11, None)
        assert rc == 0
        torch.testing.assert_close(dx.cpu(), dx_ref, rtol=1e-5, atol=2e-5)
    finally:
        lib.avc_set_tuning(b"conv_ck5", 8)


@pytest.mark.parametrize("kind", KINDS)
def test_conv_dgrad_join_and_mask(kind):
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(11)
    B, C, T = 2, 32, 21
    w = torch.randn(C, C, 5, generator=g) / 12
    dy = torch.randn(B, C, T, generator=g)
    gnext = torch.randn(B, C, (T + 1) // 2, generator=g)
    a_prev = torch.randn(B, C, T, generator=g)
    x = torch.randn(B, C, T, generator=g, requires_grad=True)
    y = O.pad_conv(x, w, None, 1)
    pooled = O.avg_pool_ceil(x, 2)
    (ref,) = torch.autograd.grad([y, pooled], x, [dy, gnext])
    wpd = pack(lib, dev, [w.to(dev)], 1)
    dx = torch.full((B, C, T), float("nan"), device=dev)
    dx2 = torch.full((B, C, T), float("nan"), device=dev)
    dyd, gd, ad = dy.to(dev), gnext.to(dev), a_prev.to(dev)
    rc = lib.avc_conv1d_dgrad(P(dyd), dyd.stride(0), dyd.stride(1), 1, 1, B, C, T, P(wpd), C, 5, 1, T, P(dx), dx.stride(0),
                              dx.stride(1), 1, P(gd), 3, gd.stride(0), gd.stride(1), 1, gd.shape[2], P(dx2), P(ad), 11, None)
    assert rc == 0
    torch.testing.assert_close(dx.cpu(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dx2.cpu(), ref * (a_prev > 0), rtol=1e-5, atol=1e-5)


WG = [
    # B, Cin, Cout, T, KS, stride
    (2, 16, 32, 32, 5, 1),
    (3, 24, 40, 20, 5, 2),
    (2, 70, 33, 21, 5, 2),
    (1, 16, 32, 70, 5, 1),
    (2, 8, 32, 19, 8, 1),
    (2, 8, 32, 17, 2, 1),
    (1, 40, 32, 33, 1, 1),
    (9, 16, 32, 3, 5, 1),
    (1, 16, 130, 200, 3, 1),
    (2, 130, 40, 40, 1, 1),      # 1x1 with the 128x128 four-accumulator tile
    (2, 80, 32, 33, 4, 1),       # Cin = 80: 128co x 32ci tile
    (8, 16, 32, 16, 5, 1),       # whole short samples per chunk (precomputed descriptors incl. reflection)
    (8, 16, 32, 32, 5, 2),
    (16, 8, 32, 8, 5, 1),
    (7, 16, 32, 16, 5, 1),       # ... B not a multiple of samples-per-chunk: generic path
    (4, 16, 32, 64, 5, 1),       # whole 32-column chunks of longer samples
    (2, 64, 128, 64, 5, 1),
    (1, 128, 256, 96, 5, 1),
    pytest.param(16, 128, 128, 128, 5, 1, marks=GPU),
    pytest.param(16, 128, 128, 128, 5, 2, marks=GPU),
    pytest.param(64, 128, 256, 16, 5, 1, marks=GPU),
    pytest.param(8, 1104, 128, 128, 1, 1, marks=GPU),
    pytest.param(8, 80, 128, 128, 8, 1, marks=GPU),
    pytest.param(2, 128, 128, 1024, 5, 1, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride", WG)
def test_conv_wgrad_matches_autograd(kind, B, Cin, Cout, T, KS, stride):
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 31 + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = (torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = O.pad_conv(x, w, b, stride)
    dy = torch.randn(y.shape, generator=g)
    dw_ref, db_ref = torch.autograd.grad(y, [w, b], dy)
    ws = torch.full((lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, y.shape[2], KS),), float("nan"), device=dev)
    dW = torch.full((Cout, Cin, KS), float("nan"), device=dev)
    db = torch.full((Cout,), float("nan"), device=dev)
    xd, dyd = x.to(dev), dy.to(dev)
    rc = lib.avc_conv1d_wgrad(P(xd), xd.stride(0), xd.stride(1), 1, P(dyd), dyd.stride(0), dyd.stride(1), 1, 1, B, Cin, Cout,
                              T, y.shape[2], KS, stride, P(dW), P(db), P(ws), None)
    assert rc == 0
    scale = dw_ref.abs().max().item()
    torch.testing.assert_close(dW.cpu(), dw_ref, rtol=1e-4, atol=1e-5 * max(1.0, scale))
    torch.testing.assert_close(db.cpu(), db_ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS", [(2, 64, 128, 64, 5), (1, 128, 256, 96, 5), (3, 64, 128, 40, 5),
                                             pytest.param(16, 128, 128, 128, 5, marks=GPU), pytest.param(4, 128, 256, 1024, 5, marks=GPU)])
def test_conv_wgrad_eight_consumer_waves(kind, B, Cin, Cout, T, KS):
    """avc_set_tuning("wgrad_cw8", 1) (opt-in): the k = 5 layers with Cin % 64 == 0 and Cout % 128 == 0 run on 128 co x 64 ci tiles with
    eight consumer waves + four producers (768 threads).  Same bar as the default 64 x 64 instance."""
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    assert lib.avc_set_tuning(b"wgrad_cw8", 1) == 0
    try:
        test_conv_wgrad_matches_autograd(kind, B, Cin, Cout, T, KS, 1)
    finally:
        lib.avc_set_tuning(b"wgrad_cw8", 0)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride", [(20, 16, 32, 5, 5, 2), (40, 8, 32, 5, 8, 1), (24, 16, 32, 7, 7, 2), (33, 16, 32, 1, 1, 1)])
def test_conv_wgrad_many_short_samples_per_chunk(kind, B, Cin, Cout, T, KS, stride):
    """T_l of a few frames (e.g. T=24 inputs reach T_l=3): many samples share a 32-column
    K-chunk and the X tile is staged without registers."""
    test_conv_wgrad_matches_autograd(kind, B, Cin, Cout, T, KS, stride)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS", [(2, 16, 32, 32, 5), (3, 64, 64, 64, 5), (2, 80, 32, 96, 5), (1, 16, 130, 200, 3), (2, 130, 40, 40, 1),
                                             (2, 80, 32, 33, 4), (2, 16, 32, 70, 6), (2, 16, 32, 64, 2), (2, 16, 32, 64, 7), (2, 8, 32, 19, 8),
                                             pytest.param(64, 128, 128, 128, 5, marks=GPU), pytest.param(256, 128, 128, 32, 5, marks=GPU),
                                             pytest.param(16, 1104, 128, 128, 1, marks=GPU), pytest.param(32, 80, 128, 128, 6, marks=GPU),
                                             pytest.param(32, 80, 128, 128, 8, marks=GPU)])
def test_conv_wgrad_x3_split_bf16_products(kind, B, Cin, Cout, T, KS):
    """avc_set_tuning("wgrad_x3", 1): the whole-chunk k = 5 weight-gradient launches form their products from three bf16 terms
    per operand (both operands are activations: both are split in registers).  Same bar as the exact-fp32 kernel, plus the
    distance from an fp64 reference next to the fp32 autograd result's own."""
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    assert lib.avc_set_tuning(b"wgrad_x3", 1) == 0
    try:
        test_conv_wgrad_matches_autograd(kind, B, Cin, Cout, T, KS, 1)
        g = torch.Generator().manual_seed(B + T)
        x = torch.randn(B, Cin, T, generator=g)
        w = (torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5)
        w64 = w.double().requires_grad_(True)
        y64 = O.pad_conv(x.double(), w64, None, 1)
        dy = torch.randn(y64.shape, generator=g)
        (dw64,) = torch.autograd.grad(y64, [w64], dy.double())
        w32 = w.clone().requires_grad_(True)
        (dw32,) = torch.autograd.grad(O.pad_conv(x, w32, None, 1), [w32], dy)
        To = y64.shape[2]
        ws = torch.zeros(lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, To, KS), device=dev)
        dW = torch.zeros(Cout, Cin, KS, device=dev)
        db = torch.zeros(Cout, device=dev)
        xd, dyd = x.to(dev), dy.to(dev)
        assert lib.avc_conv1d_wgrad(P(xd), xd.stride(0), xd.stride(1), 1, P(dyd), dyd.stride(0), dyd.stride(1), 1, 1, B, Cin, Cout, T, To, KS, 1,
                                    P(dW), P(db), P(ws), None) == 0
        e_x3 = (dW.cpu().double() - dw64).abs().max().item()
        e_32 = (dw32.double() - dw64).abs().max().item()
        print(f"[{kind} wgrad x3 B={B} {Cin}->{Cout} T={T} k={KS}] max |err| vs fp64: split-bf16 {e_x3:.2e}, fp32 autograd {e_32:.2e}")
        assert e_x3 <= (4.0 if kind == "gpu" else 12.0) * e_32 + 1e-6 * dw64.abs().max().item()
    finally:
        lib.avc_set_tuning(b"wgrad_x3", 0)


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride", [(2, 40, 32, 32, 5, 1), (3, 16, 32, 20, 5, 2), (2, 24, 32, 33, 1, 1),
                                                     pytest.param(8, 128, 128, 128, 5, 1, marks=GPU),
                                                     pytest.param(64, 128, 128, 16, 5, 1, marks=GPU)])
def test_conv_bf16_operand_mode(kind, B, Cin, Cout, T, KS, stride):
    """avc_set_tuning("compute", 1): operands rounded to bf16 (RNE) inside the matrix core, fp32
    accumulate.  bf16 x bf16 products are exact in fp32, so forward and wgrad must equal the fp32
    ops applied to bf16-rounded operands up to summation order; dgrad rounds the reflect-folded
    gradient, so it is compared with the fp32 result at bf16 accuracy."""
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(7 * B + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    lib.avc_set_tuning(b"compute", 1)
    try:
        out, _ = conv_fwd(lib, dev, x.to(dev), w.to(dev), b.to(dev), stride=stride, act=1)
        ref = torch.relu(O.pad_conv(bf16r(x), bf16r(w), b, stride))
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)
        assert (out.cpu() - torch.relu(O.pad_conv(x, w, b, stride))).abs().max() > 1e-4  # really a different precision
        # wgrad
        xr, wr = x.clone(), w.clone().requires_grad_(True)
        y = O.pad_conv(bf16r(xr), wr, None, stride)
        dy = torch.randn(y.shape, generator=g)
        (dw_ref,) = torch.autograd.grad(y, [wr], bf16r(dy))
        To = y.shape[2]
        ws = torch.zeros(lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, To, KS), device=dev)
        dW = torch.zeros(Cout, Cin, KS, device=dev)
        db = torch.zeros(Cout, device=dev)
        xd, dyd = x.to(dev), dy.to(dev)
        assert lib.avc_conv1d_wgrad(P(xd), xd.stride(0), xd.stride(1), 1, P(dyd), dyd.stride(0), dyd.stride(1), 1, 1, B, Cin, Cout,
                                    T, To, KS, stride, P(dW), P(db), P(ws), None) == 0
        torch.testing.assert_close(dW.cpu(), dw_ref, rtol=1e-4, atol=1e-5 * max(1.0, dw_ref.abs().max().item()))
        torch.testing.assert_close(db.cpu(), dy.sum((0, 2)), rtol=1e-4, atol=1e-4)   # the bias gradient stays fp32
        # dgrad
        xg = x.clone().requires_grad_(True)
        (dx_ref,) = torch.autograd.grad(O.pad_conv(xg, w, None, stride), [xg], dy)
        wpd = pack(lib, dev, [w.to(dev)], 1)
        dx = torch.zeros(B, Cin, T, device=dev)
        assert lib.avc_conv1d_dgrad(P(dyd), dyd.stride(0), dyd.stride(1), 1, 1, B, Cout, To, P(wpd), Cin, KS, stride, T, P(dx),
                                    dx.stride(0), dx.stride(1), 1, None, 0, 0, 0, 0, 0, None, None, 0, None) == 0
        err = ((dx.cpu() - dx_ref).norm() / dx_ref.norm()).item()
        assert 1e-5 < err < 1e-2, err
    finally:
        lib.avc_set_tuning(b"compute", 0)


# ---- split-bf16 conv kernel (csrc/conv_x3.hip): tile code 97 + its own weight image.  Every operand as three bf16 terms,
# six bf16 MFMAs per product block: the result must be as close to exact arithmetic as an fp32 convolution is.
def pack_x3(lib, dev, w, dgrad):
    Cout, Cin, KS = w.shape
    n = lib.avc_packed_weight_floats_x3(Cout, Cin, KS, dgrad)
    assert n > 0
    dst = torch.zeros(n, device=dev)
    assert lib.avc_pack_weight_x3(P(w), Cout, Cin, KS, dgrad, P(dst), None) == 0
    return dst


X3 = [
    # B, Cin, Cout, T, stride   (k = 5; a negative stride marks a 1x1 conv)
    (2, 40, 48, 40, -1),      # 1x1: two 32-channel chunks, the second zero-padded (40 channels)
    (1, 80, 130, 70, -1),     # 1x1: three chunks (80 = 2.5), three row tiles
    (2, 16, 32, 40, 1),       # one chunk, partial second tile
    (3, 32, 48, 16, 1),       # two chunks, several samples per tile, rows >= Cout masked
    (2, 32, 32, 21, 2),       # stride 2, odd T (dgrad: all taps)
    (2, 16, 32, 32, 2),       # stride 2 (dgrad: one column parity per wave)
    (1, 32, 144, 70, 1),      # three row tiles
    pytest.param(64, 128, 128, 128, 1, marks=GPU),
    pytest.param(64, 128, 128, 64, 2, marks=GPU),
    pytest.param(64, 1104, 128, 128, -1, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,stride", X3)
def test_conv_x3_fwd_and_dgrad_are_fp32_accurate(kind, B, Cin, Cout, T, stride):
    if kind == "emu" and B * Cin * Cout * T > 3e6:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    KS = 5
    if stride < 0:
        KS, stride = 1, 1
    g = torch.Generator().manual_seed(B * 31 + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    xd64 = x.double().requires_grad_(True)
    y64 = O.pad_conv(xd64, w.double(), b.double(), stride)
    y32 = O.pad_conv(x, w, b, stride)
    wp = pack_x3(lib, dev, w.to(dev), 0)
    out = torch.full(y32.shape, float("nan"), device=dev)
    xd = x.to(dev)
    rc = lib.avc_conv1d_fwd(P(xd), xd.stride(0), xd.stride(1), 1, B, Cin, T, P(wp), P(b.to(dev)), Cout, KS, stride, 0, P(out),
                            out.stride(0), out.stride(1), 1, 1, None, 0, 0, 0, 0, 0, None, 97, None)
    assert rc == 0, rc
    e_x3 = (out.cpu().double() - y64.detach()).abs().max().item()
    e_32 = (y32.double() - y64.detach()).abs().max().item()
    print(f"[{kind} x3 fwd B={B} {Cin}->{Cout} T={T} s={stride}] max |err| vs fp64: split-bf16 {e_x3:.2e}, fp32 conv {e_32:.2e}")
    # the hardware sums the 16 products of an instruction before it rounds into the accumulator; the CPU simulator rounds after
    # every product (6 x more roundings than an fp32 convolution), so its bar is wider
    bar = 4.0 if kind == "gpu" else 10.0
    assert e_x3 <= bar * e_32 + 1e-7, (e_x3, e_32)
    torch.testing.assert_close(out.cpu(), y32, rtol=1e-5, atol=2e-5)
    # input gradient
    dy = torch.randn(y32.shape, generator=g)
    (dx64,) = torch.autograd.grad(y64, xd64, dy.double())
    xg = x.clone().requires_grad_(True)
    (dx32,) = torch.autograd.grad(O.pad_conv(xg, w, None, stride), xg, dy)
    if KS == 1 and Cout % 16 != 0 and Cout < 32:
        return
    wpd = pack_x3(lib, dev, w.to(dev), 1)
    dx = torch.full((B, Cin, T), float("nan"), device=dev)
    dyd = dy.to(dev)
    rc = lib.avc_conv1d_dgrad(P(dyd), dyd.stride(0), dyd.stride(1), 1, 1, B, Cout, dy.shape[2], P(wpd), Cin, KS, stride, T, P(dx),
                              dx.stride(0), dx.stride(1), 1, None, 0, 0, 0, 0, 0, None, None, 97, None)
    if T < 10 and KS == 5:
        assert rc != 0
        return
    assert rc == 0, rc
    e_x3 = (dx.cpu().double() - dx64).abs().max().item()
    e_32 = (dx32.double() - dx64).abs().max().item()
    print(f"[{kind} x3 dgrad] max |err| vs fp64: split-bf16 {e_x3:.2e}, fp32 conv {e_32:.2e}")
    assert e_x3 <= bar * e_32 + 1e-7, (e_x3, e_32)
    torch.testing.assert_close(dx.cpu(), dx32, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride", [(4, 16, 32, 64, 5, 1), (6, 80, 32, 64, 7, 1), (8, 16, 32, 32, 5, 2), (3, 130, 40, 96, 1, 1), (4, 64, 128, 64, 5, 1)])
def test_conv_wgrad_reduce_is_order_independent(B, Cin, Cout, T, KS, stride, monkeypatch):
    """The stream-K launch leaves partial tiles in slots; the reduce launch sums every tile's slots in a FIXED order -- so the gradient
    must be bit-identical whatever order the workgroups of either launch ran in.  The simulator runs them in ascending, reversed and
    interleaved order (hardware promises none)."""
    lib, dev = backend("emu")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, T, generator=g)
    To = (T + 2 * (KS // 2) - (1 if KS % 2 == 0 else 0) - KS) // stride + 1
    dy = torch.randn(B, Cout, To, generator=g)
    res = []
    for order in ("", "reverse", "stride"):
        monkeypatch.setenv("AVC_EMU_BLOCK_ORDER", order)
        ws = torch.full((lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, To, KS),), float("nan"))
        dW = torch.full((Cout, Cin, KS), float("nan"))
        db = torch.full((Cout,), float("nan"))
        assert lib.avc_conv1d_wgrad(P(x), x.stride(0), x.stride(1), 1, P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cin, Cout, T, To, KS, stride,
                                    P(dW), P(db), P(ws), None) == 0
        res.append((dW, db))
    w = (torch.randn(Cout, Cin, KS, generator=g)).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    dw_ref, db_ref = torch.autograd.grad(O.pad_conv(x, w, b, stride), [w, b], dy)
    torch.testing.assert_close(res[0][0], dw_ref, rtol=1e-4, atol=1e-4)
    for dW, db in res[1:]:
        assert torch.equal(dW, res[0][0]) and torch.equal(db, res[0][1])
