"""Kernel-logic tests of the implicit-GEMM conv (forward + dgrad) on the CPU
lane-level simulator, against torch CPU ops restating model.py:21-32."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from oracle import avc_oracle as O
from tests.emu_util import I, L, P, emu_lib


def pack(lib, ws, dgrad):
    Cout, Cin, KS = ws[0].shape[0] * len(ws), ws[0].shape[1], (ws[0].shape[2] if ws[0].dim() == 3 else 1)
    lib.avc_packed_weight_floats.restype = ctypes.c_long
    n = lib.avc_packed_weight_floats(Cout, Cin, KS, dgrad)
    dst = torch.full((n,), float("nan"))
    arr = (ctypes.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    rc = lib.avc_pack_weight(arr, len(ws), ws[0].shape[0], Cout, Cin, KS, dgrad, P(dst), None)
    assert rc == 0
    assert torch.isfinite(dst).all()
    return dst


def conv_fwd(lib, x, w, b, stride=1, act=0, tile=0, res=None, res_mode=0, ops=1):
    B, Cin, Tin = x.shape
    Cout, _, KS = w.shape
    wp = pack(lib, [w], 0)
    padL, padR = KS // 2, (KS // 2 - 1 if KS % 2 == 0 else KS // 2)
    Tout = (Tin + padL + padR - KS) // stride + 1
    out = torch.full((B, Cout // ops, Tout * ops), float("nan"))
    out2 = torch.full_like(out, float("nan")) if res is not None else None
    rb = rc = rt = Tres = 0
    if res is not None:
        rb, rc, rt, Tres = res.stride(0), res.stride(1), res.stride(2), res.shape[2]
    rcode = lib.avc_conv1d_fwd(P(x), L(x.stride(0)), L(x.stride(1)), I(x.stride(2)), B, Cin, Tin, P(wp), P(b), Cout, KS,
                               stride, act, P(out), L(out.stride(0)), L(out.stride(1)), I(out.stride(2)), ops, P(res),
                               res_mode, L(rb), L(rc), I(rt), Tres, P(out2), tile, None)
    assert rcode == 0, rcode
    return out, out2


CASES = [
    # B, Cin, Cout, T, KS, stride, tile
    (2, 16, 32, 32, 5, 1, 11),
    (3, 24, 40, 20, 5, 2, 11),
    (1, 16, 32, 70, 5, 1, 11),   # Tout > BN, partial last tile
    (2, 8, 32, 19, 8, 1, 11),    # even kernel (bank), odd T
    (2, 8, 32, 17, 2, 1, 11),
    (1, 40, 32, 33, 1, 1, 11),   # 1x1
    (2, 16, 130, 64, 3, 1, 21),  # 128x64 tile, 2 M tiles
    (1, 16, 128, 130, 5, 1, 22),  # 128x128 tile
    (5, 16, 32, 3, 5, 1, 11),    # bottleneck-sized rows
    (2, 16, 32, 21, 5, 2, 11),   # stride 2, odd T
]


@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,tile", CASES)
def test_conv_fwd_matches_pad_conv(B, Cin, Cout, T, KS, stride, tile):
    lib = emu_lib()
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = torch.relu(O.pad_conv(x, w, b, stride))
    out, _ = conv_fwd(lib, x, w, b, stride, act=1, tile=tile)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


def test_conv_fwd_transposed_input_view_and_residual_pool():
    lib = emu_lib()
    g = torch.Generator().manual_seed(7)
    B, Cin, Cout, T = 2, 16, 32, 22
    xt = torch.randn(B, T, Cin, generator=g)
    x = xt.transpose(1, 2)  # the [B,M,T] view of data_utils.py:14-16 (strides (T*M, 1, M))
    w = torch.randn(Cout, Cin, 5, generator=g) / 9
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    y = torch.relu(O.pad_conv(x, w, b, 2))
    out, out2 = conv_fwd(lib, x, w, b, 2, act=1, tile=11, res=res, res_mode=2)
    torch.testing.assert_close(out, y, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out2, y + O.avg_pool_ceil(res, 2), rtol=1e-5, atol=1e-5)


def test_conv_fwd_pixel_shuffle_store():
    lib = emu_lib()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 16, 12, generator=g)
    w = torch.randn(64, 16, 5, generator=g) / 9
    b = torch.randn(64, generator=g)
    ref = O.pixel_shuffle_1d(O.pad_conv(x, w, b), 2)
    out, _ = conv_fwd(lib, x, w, b, 1, act=0, tile=11, ops=2)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


DG = [
    (2, 16, 32, 32, 5, 1, 11),
    (3, 24, 40, 20, 5, 2, 11),
    (2, 16, 32, 21, 5, 2, 11),
    (1, 16, 32, 70, 5, 1, 11),
    (1, 16, 32, 130, 5, 1, 11),  # right mirror spans the last two tiles
    (1, 40, 32, 33, 1, 1, 11),
    (5, 16, 32, 3, 5, 1, 11),
    (1, 16, 128, 130, 5, 1, 22),
    (2, 16, 32, 66, 5, 1, 21),
]


@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,tile", DG)
def test_conv_dgrad_matches_autograd(B, Cin, Cout, T, KS, stride, tile):
    lib = emu_lib()
    g = torch.Generator().manual_seed(B * 77 + T)
    x = torch.randn(B, Cin, T, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    y = O.pad_conv(x, w, None, stride)
    dy = torch.randn(y.shape, generator=g)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    wpd = pack(lib, [w], 1)
    dx = torch.full((B, Cin, T), float("nan"))
    rc = lib.avc_conv1d_dgrad(P(dy), L(dy.stride(0)), L(dy.stride(1)), I(dy.stride(2)), 1, B, Cout, dy.shape[2], P(wpd),
                              Cin, KS, stride, T, P(dx), L(dx.stride(0)), L(dx.stride(1)), I(dx.stride(2)), None, 0,
                              L(0), L(0), I(0), 0, None, None, tile, None)
    assert rc == 0
    torch.testing.assert_close(dx, dx_ref, rtol=1e-5, atol=1e-5)


def test_conv_dgrad_join_and_mask():
    lib = emu_lib()
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
    wpd = pack(lib, [w], 1)
    dx = torch.full((B, C, T), float("nan"))
    dx2 = torch.full((B, C, T), float("nan"))
    rc = lib.avc_conv1d_dgrad(P(dy), L(dy.stride(0)), L(dy.stride(1)), I(1), 1, B, C, T, P(wpd), C, 5, 1, T, P(dx),
                              L(dx.stride(0)), L(dx.stride(1)), I(1), P(gnext), 3, L(gnext.stride(0)),
                              L(gnext.stride(1)), I(1), gnext.shape[2], P(dx2), P(a_prev), 11, None)
    assert rc == 0
    torch.testing.assert_close(dx, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dx2, ref * (a_prev > 0), rtol=1e-5, atol=1e-5)


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
]


@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride", WG)
def test_conv_wgrad_matches_autograd(B, Cin, Cout, T, KS, stride):
    lib = emu_lib()
    g = torch.Generator().manual_seed(B * 31 + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = (torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = O.pad_conv(x, w, b, stride)
    dy = torch.randn(y.shape, generator=g)
    dw_ref, db_ref = torch.autograd.grad(y, [w, b], dy)
    lib.avc_conv1d_wgrad_ws_floats.restype = ctypes.c_long
    ws = torch.full((lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, y.shape[2], KS),), float("nan"))
    dW = torch.full((Cout, Cin, KS), float("nan"))
    db = torch.full((Cout,), float("nan"))
    rc = lib.avc_conv1d_wgrad(P(x), L(x.stride(0)), L(x.stride(1)), I(1), P(dy), L(dy.stride(0)), L(dy.stride(1)), I(1),
                              1, B, Cin, Cout, T, y.shape[2], KS, stride, P(dW), P(db), P(ws), None)
    assert rc == 0
    torch.testing.assert_close(dW, dw_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db, db_ref, rtol=1e-4, atol=1e-4)
