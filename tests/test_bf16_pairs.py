"""bf16 PAIR storage kernels (compute_dtype "bf16": activations [B, C, T] stored as dwords [B][C/2][T] of two bf16 channels,
adaptive_voice_conversion_amd/csrc/bf16_pairs.h) vs torch restatements evaluated on the same bf16-rounded operands.

Reference semantics: model.py:21-32 (pad_layer + Conv1d), :296,341 (InstanceNorm1d), :77-83 (append_cond), :52-63 (pixel
shuffle / nearest upsample), :248,319 (ceil-mode avg-pool) and their autograd.  Tolerance: the kernels accumulate in fp32 and
round ONCE on store, so a stored value is the bf16 rounding of an fp32 result that differs from the reference's only by
summation order: |err| <= 1 bf16 ulp (2^-8 relative) + a small absolute term."""
import ctypes

import pytest
import torch

from oracle import avc_oracle as O
from tests.emu_util import KINDS, P, backend

GPU = pytest.mark.gpu


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def to_pairs(x):
    """[B, C, T] fp32 -> int32 [B, C/2, T]: low half = bf16(channel 2p), high half = bf16(channel 2p+1)"""
    assert x.shape[1] % 2 == 0
    u = x.to(torch.bfloat16).contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    return (u[:, 0::2] | (u[:, 1::2] << 16)).contiguous()


def from_pairs(p):
    lo = (p & 0xFFFF).to(torch.int16).view(torch.bfloat16).to(torch.float32)
    hi = ((p >> 16) & 0xFFFF).to(torch.int16).view(torch.bfloat16).to(torch.float32)
    B, C2, T = p.shape
    out = torch.empty(B, 2 * C2, T)
    out[:, 0::2] = lo
    out[:, 1::2] = hi
    return out


def close_bf16(got, ref, ulps=1.0, atol=1e-3):
    err = (got - ref).abs()
    bound = ulps * 2.0 ** -8 * ref.abs() + atol
    bad = err > bound
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} off; worst err {err.max().item():.3e} at ref {ref.flatten()[err.argmax()].item():.3e}"


class op_dtype:
    def __init__(self, lib, d):
        self.lib, self.d = lib, d

    def __enter__(self):
        assert self.lib.avc_set_tuning(b"op_compute_dtype", self.d) == 0

    def __exit__(self, *a):
        self.lib.avc_set_tuning(b"op_compute_dtype", 0)


def pack(lib, dev, w, dgrad):
    Cout, Cin, KS = w.shape
    n = lib.avc_packed_weight_floats(Cout, Cin, KS, dgrad)
    dst = torch.zeros(n, device=dev)
    arr = (ctypes.c_void_p * 1)(w.data_ptr())
    assert lib.avc_pack_weight(arr, 1, Cout, Cout, Cin, KS, dgrad, P(dst), None) == 0
    return dst


def test_pair_helpers_round_trip():
    x = torch.randn(2, 6, 5)
    torch.testing.assert_close(from_pairs(to_pairs(x)), bf16r(x), rtol=0, atol=0)


FWD = [
    # B, Cin, Cout, T, KS, stride, tile
    (2, 32, 32, 32, 5, 1, 11),
    (3, 48, 40, 20, 5, 2, 11),
    (1, 32, 32, 70, 5, 1, 11),
    (2, 16, 32, 19, 8, 1, 11),
    (1, 80, 32, 33, 1, 1, 11),
    (2, 32, 130, 64, 3, 1, 21),
    (2, 40, 32, 32, 5, 1, 11),     # 20 dword channels: the last unit straddles
    (5, 32, 32, 3, 5, 1, 11),
    (3, 128, 64, 16, 5, 1, 11),    # several short samples per tile, split-K groups
    (5, 256, 40, 16, 5, 1, 0),
    (9, 128, 128, 32, 5, 1, 0),
    pytest.param(8, 128, 128, 128, 5, 1, 0, marks=GPU),
    pytest.param(8, 128, 256, 64, 5, 1, 0, marks=GPU),
    pytest.param(64, 128, 128, 16, 5, 1, 0, marks=GPU),
    pytest.param(4, 1104, 128, 128, 1, 1, 0, marks=GPU),
    pytest.param(4, 80, 128, 128, 8, 1, 21, marks=GPU),
    pytest.param(2, 128, 128, 1024, 5, 1, 21, marks=GPU),
]


def _fwd_cases():
    """FWD x {pair output, fp32 output}; the fp32-output epilogue is covered by ONE tile shape (11) -- the other combinations are not
    generated (they used to be skips in the driver's log)."""
    out = []
    for e in FWD:
        vals = tuple(e.values) if hasattr(e, "values") else tuple(e)
        marks = tuple(e.marks) if hasattr(e, "marks") else ()
        for f32out in (False, True):
            if f32out and vals[6] != 11:
                continue
            out.append(pytest.param(*vals, f32out, marks=marks))
    return out


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,tile,f32out", _fwd_cases())
def test_pairs_conv_fwd(kind, B, Cin, Cout, T, KS, stride, tile, f32out):
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = torch.relu(O.pad_conv(bf16r(x).double(), bf16r(w).double(), b.double(), stride)).float()
    xp = to_pairs(x).to(dev)
    padL, padR = KS // 2, (KS // 2 - 1 if KS % 2 == 0 else KS // 2)
    Tout = (T + padL + padR - KS) // stride + 1
    with op_dtype(lib, 4 if f32out else 3):
        wp = pack(lib, dev, w.to(dev), 0)
        bd = b.to(dev)
        if f32out:
            out = torch.full((B, Cout, Tout), float("nan"), device=dev)
        else:
            out = torch.zeros(B, Cout // 2, Tout, dtype=torch.int32, device=dev)
        rc = lib.avc_conv1d_fwd(P(xp), xp.stride(0), xp.stride(1), 1, B, Cin, T, P(wp), P(bd), Cout, KS, stride, 1, P(out), out.stride(0),
                                out.stride(1), 1, 1, None, 0, 0, 0, 0, 0, None, tile, None)
    assert rc == 0, rc
    if f32out:
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    else:
        close_bf16(from_pairs(out.cpu()), ref)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("res_mode,stride", [(1, 1), (2, 2)])
def test_pairs_conv_fwd_residual_second_output(kind, res_mode, stride):
    """speaker-encoder block tail (model.py:241-249): a2 = relu(conv(a1)), out = a2 + [avg-pooled] residual, both stored as pairs"""
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(7 + res_mode)
    B, C, T = 2, 32, 22
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(C, C, 5, generator=g) / 12
    b = torch.randn(C, generator=g)
    res = torch.randn(B, C, T, generator=g)
    y = torch.relu(O.pad_conv(bf16r(x).double(), bf16r(w).double(), b.double(), stride)).float()
    r = bf16r(res)
    ref2 = y + (O.avg_pool_ceil(r, 2) if res_mode == 2 else r)
    Tout = y.shape[2]
    xp, rp = to_pairs(x).to(dev), to_pairs(res).to(dev)
    out = torch.zeros(B, C // 2, Tout, dtype=torch.int32, device=dev)
    out2 = torch.zeros_like(out)
    with op_dtype(lib, 3):
        wp = pack(lib, dev, w.to(dev), 0)
        bd = b.to(dev)
        rc = lib.avc_conv1d_fwd(P(xp), xp.stride(0), xp.stride(1), 1, B, C, T, P(wp), P(bd), C, 5, stride, 1, P(out), out.stride(0), out.stride(1), 1, 1,
                                P(rp), res_mode, rp.stride(0), rp.stride(1), 1, T, P(out2), 11, None)
    assert rc == 0, rc
    close_bf16(from_pairs(out.cpu()), y)
    close_bf16(from_pairs(out2.cpu()), ref2)


DG = [
    (2, 32, 32, 32, 5, 1, 11),
    (3, 48, 40, 20, 5, 2, 11),
    (2, 32, 32, 21, 5, 2, 11),
    (1, 32, 32, 130, 5, 1, 11),
    (1, 80, 32, 33, 1, 1, 11),
    (5, 32, 32, 3, 5, 1, 11),     # very short samples: both mirror windows per column
    (1, 32, 128, 130, 5, 1, 21),
    (2, 40, 36, 32, 5, 1, 11),
    (5, 40, 128, 16, 5, 1, 0),
    (3, 32, 32, 32, 5, 2, 11),    # stride 2, one column parity per wave
    (1, 32, 32, 101, 5, 2, 11),
    (2, 64, 64, 32, 5, 2, 11),
    (3, 128, 128, 32, 5, 1, 0),
    pytest.param(8, 128, 128, 128, 5, 1, 0, marks=GPU),
    pytest.param(8, 128, 128, 128, 5, 2, 0, marks=GPU),
    pytest.param(64, 128, 128, 16, 5, 1, 0, marks=GPU),
    pytest.param(2, 128, 128, 1024, 5, 1, 0, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,tile", DG)
def test_pairs_conv_dgrad(kind, B, Cin, Cout, T, KS, stride, tile):
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 77 + T)
    x = torch.randn(B, Cin, T, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    y = O.pad_conv(x, bf16r(w).double(), None, stride)
    dy = torch.randn(y.shape, generator=g)
    (dx_ref,) = torch.autograd.grad(y, x, bf16r(dy).double())
    dyp = to_pairs(dy).to(dev)
    dx = torch.zeros(B, Cin // 2, T, dtype=torch.int32, device=dev)
    with op_dtype(lib, 3):
        wpd = pack(lib, dev, w.to(dev), 1)
        rc = lib.avc_conv1d_dgrad(P(dyp), dyp.stride(0), dyp.stride(1), 1, 1, B, Cout, dy.shape[2], P(wpd), Cin, KS, stride, T, P(dx), dx.stride(0),
                                  dx.stride(1), 1, None, 0, 0, 0, 0, 0, None, None, tile, None)
    assert rc == 0, rc
    close_bf16(from_pairs(dx.cpu()), dx_ref.float(), atol=2e-3)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("res_mode", [1, 3, 4])
def test_pairs_conv_dgrad_join_and_mask(kind, res_mode):
    """dx = dgrad(dy) + adjoint-of-the-residual-path(g_next); dx2 = dx * (a_prev > 0)  (the block backward of model.py:241-249 / :353-369)"""
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(11 + res_mode)
    B, C, T = 2, 32, 21 if res_mode == 3 else 20
    w = torch.randn(C, C, 5, generator=g) / 12
    dy = torch.randn(B, C, T, generator=g)
    Tn = {1: T, 3: (T + 1) // 2, 4: 2 * T}[res_mode]
    gnext = torch.randn(B, C, Tn, generator=g)
    a_prev = torch.randn(B, C, T, generator=g)
    x = torch.randn(B, C, T, generator=g, dtype=torch.float64, requires_grad=True)
    y = O.pad_conv(x, bf16r(w).double(), None, 1)
    side = {1: x, 3: O.avg_pool_ceil(x, 2), 4: x.repeat_interleave(2, dim=2)}[res_mode]
    (ref,) = torch.autograd.grad([y, side], x, [bf16r(dy).double(), bf16r(gnext).double()])
    ref = ref.float()
    dyp, gp, ap = to_pairs(dy).to(dev), to_pairs(gnext).to(dev), to_pairs(a_prev).to(dev)
    dx = torch.zeros(B, C // 2, T, dtype=torch.int32, device=dev)
    dx2 = torch.zeros_like(dx)
    with op_dtype(lib, 3):
        wpd = pack(lib, dev, w.to(dev), 1)
        rc = lib.avc_conv1d_dgrad(P(dyp), dyp.stride(0), dyp.stride(1), 1, 1, B, C, T, P(wpd), C, 5, 1, T, P(dx), dx.stride(0), dx.stride(1), 1,
                                  P(gp), res_mode, gp.stride(0), gp.stride(1), 1, Tn, P(dx2), P(ap), 11, None)
    assert rc == 0, rc
    close_bf16(from_pairs(dx.cpu()), ref, atol=2e-3)
    close_bf16(from_pairs(dx2.cpu()), ref * (bf16r(a_prev) > 0), atol=2e-3)


# ---------------------------------------------------------------------------------------------------------------
# InstanceNorm / AdaIN rows
# ---------------------------------------------------------------------------------------------------------------
def to_planar(x):
    """[B, C, T] fp32 -> int32 [B, C, T/2]: the natural bf16 row of a channel, two frames per dword"""
    u = x.to(torch.bfloat16).contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    return (u[:, :, 0::2] | (u[:, :, 1::2] << 16)).contiguous()


def from_planar(p):
    lo = (p & 0xFFFF).to(torch.int16).view(torch.bfloat16).to(torch.float32)
    hi = ((p >> 16) & 0xFFFF).to(torch.int16).view(torch.bfloat16).to(torch.float32)
    B, C, T2 = p.shape
    out = torch.empty(B, C, 2 * T2)
    out[:, :, 0::2] = lo
    out[:, :, 1::2] = hi
    return out


def ref_block(y, cond, res, res_mode):
    v = O.instance_norm(y)
    if cond is not None:
        v = O.append_cond(v, cond)
    v = torch.relu(v)
    if res is not None:
        v = v + {1: lambda r: r, 2: lambda r: O.avg_pool_ceil(r, 2), 5: lambda r: O.upsample_nearest(r, 2)}[res_mode](res)
    return v


IN_CASES = [
    # B, C, T, affine, res_mode, planar
    (3, 8, 16, True, 0, 0), (2, 8, 32, False, 1, 0), (2, 8, 64, True, 5, 1), (2, 8, 128, True, 2, 0), (1, 4, 1024, True, 1, 0),
    (2, 6, 24, True, 0, 1), (1, 4, 2048, False, 0, 0),
    pytest.param(32, 128, 128, True, 5, 1, marks=GPU), pytest.param(32, 128, 16, True, 0, 0, marks=GPU),
    pytest.param(32, 128, 64, False, 2, 0, marks=GPU), pytest.param(8, 128, 1024, True, 1, 0, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,C,T,affine,res_mode,planar", IN_CASES)
def test_pairs_instnorm_fwd_bwd(kind, B, C, T, affine, res_mode, planar):
    if kind == "emu" and B * C * T > 20000:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(T * 10 + B)
    y = bf16r(torch.randn(B, C, T, generator=g) * 2 + 0.5).requires_grad_(True)      # (what the kernel reads back from HBM)
    cond_all = torch.randn(B, 3 * 2 * C, generator=g)
    off = 2 * C
    cond = cond_all[:, off:off + 2 * C].clone().requires_grad_(True) if affine else None
    res = None
    if res_mode == 1:
        res = bf16r(torch.randn(B, C, T, generator=g))
    elif res_mode == 2:
        res = bf16r(torch.randn(B, C, 2 * T, generator=g))
    elif res_mode == 5:
        res = bf16r(torch.randn(B, C, T // 2, generator=g))
    Tres = res.shape[2] if res is not None else 0
    ref = ref_block(y, cond, res, res_mode)
    yd = (to_planar(y.detach()) if planar else to_pairs(y.detach())).to(dev)
    cd = cond_all.to(dev)
    rd = to_pairs(res).to(dev) if res is not None else None
    out = torch.zeros(B, C // 2, T, dtype=torch.int32, device=dev)
    mean = torch.full((B * C,), float("nan"), device=dev)
    rstd = torch.full((B * C,), float("nan"), device=dev)
    rc = lib.avc_instnorm_fwd_pairs(P(yd), B, C, T, P(cd if affine else None), cd.stride(0), off, 1, P(rd), res_mode, Tres, planar, P(out),
                                    P(mean), P(rstd), None)
    assert rc == 0, rc
    close_bf16(from_pairs(out.cpu()), ref.detach(), atol=2e-5)
    torch.testing.assert_close(mean.cpu().view(B, C), y.detach().mean(-1), rtol=1e-5, atol=1e-5)   # fp32 statistics
    gout = bf16r(torch.randn(B, C, T, generator=g))
    grads = torch.autograd.grad(ref, [y] + ([cond] if affine else []), gout)
    dy = torch.zeros_like(yd)
    dcond = torch.zeros(B, 3 * 2 * C, device=dev)
    gd = to_pairs(gout).to(dev)
    rc = lib.avc_instnorm_bwd_pairs(P(gd), P(yd), P(mean), P(rstd), B, C, T, P(cd if affine else None), cd.stride(0), off, 1, planar, P(dy),
                                    P(dcond if affine else None), dcond.stride(0), off, None)
    assert rc == 0, rc
    got = from_planar(dy.cpu()) if planar else from_pairs(dy.cpu())
    close_bf16(got, grads[0], atol=1e-4)
    if affine:
        torch.testing.assert_close(dcond.cpu()[:, off:off + 2 * C], grads[1], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("kind", KINDS)
def test_to_pairs_of_transposed_view(kind):
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(3)
    xt = torch.randn(2, 9, 6, generator=g).to(dev)        # [B, T, M]: data_utils.py:14-16 hands over x.transpose(1, 2)
    x = xt.transpose(1, 2)
    dst = torch.zeros(2, 3, 9, dtype=torch.int32, device=dev)
    assert lib.avc_to_pairs(P(x), x.stride(0), x.stride(1), x.stride(2), 2, 6, 9, P(dst), None) == 0
    torch.testing.assert_close(from_pairs(dst.cpu()), bf16r(x.cpu()), rtol=0, atol=0)


# ---------------------------------------------------------------------------------------------------------------
# weight gradient
# ---------------------------------------------------------------------------------------------------------------
WG = [
    # B, Cin, Cout, T, KS, stride
    (2, 32, 32, 32, 5, 1),
    (3, 24, 40, 20, 5, 2),
    (2, 70, 34, 21, 5, 2),
    (1, 16, 32, 70, 5, 1),
    (2, 8, 32, 19, 8, 1),
    (1, 40, 32, 33, 1, 1),
    (9, 16, 32, 3, 5, 1),
    (2, 130, 40, 40, 1, 1),      # 1x1 with the 128x128 four-accumulator tile
    (2, 80, 32, 64, 4, 1),       # Cin = 80: 128co x 32ci tile, whole chunks
    (8, 16, 32, 16, 5, 1),       # whole short samples per chunk
    (8, 16, 32, 32, 5, 2),
    (7, 16, 32, 16, 5, 1),
    (4, 64, 64, 64, 5, 1),       # whole 32-column chunks (ds_read_b128 path; round 6: 16-byte staged rows, first + last chunk of a sample)
    (3, 64, 64, 32, 5, 1),       # ... one chunk per sample: both mirrored edges in the same fragment set
    (2, 128, 64, 96, 5, 1),      # ... three chunks per sample (an interior chunk), two ci tiles
    (2, 64, 128, 64, 1, 1),      # ... 1x1 (no halo pieces read)
    (1, 256, 128, 64, 1, 1),     # ... 1x1 on the 128 x 128 four-accumulator tile
    (2, 80, 128, 32, 8, 1),      # ... run-time tap counts (the bank's instance): fragments read at a dword-aligned LDS address, mirrored frames by (padL, padR)
    (1, 80, 64, 64, 3, 1),
    (1, 16, 32, 96, 6, 1),
    (2, 80, 32, 64, 5, 1),
    (1, 16, 32, 64, 2, 1),
    (1, 16, 32, 64, 7, 1),
    pytest.param(16, 128, 128, 128, 5, 1, marks=GPU),
    pytest.param(16, 128, 128, 128, 5, 2, marks=GPU),
    pytest.param(64, 128, 256, 16, 5, 1, marks=GPU),
    pytest.param(8, 1104, 128, 128, 1, 1, marks=GPU),
    pytest.param(8, 80, 128, 128, 8, 1, marks=GPU),
    pytest.param(2, 128, 128, 1024, 5, 1, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride", WG)
def test_pairs_conv_wgrad(kind, B, Cin, Cout, T, KS, stride):
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 31 + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = (torch.randn(Cout, Cin, KS, generator=g, dtype=torch.float64) / (Cin * KS) ** 0.5).requires_grad_(True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = O.pad_conv(bf16r(x).double(), w, b, stride)
    dy = torch.randn(y.shape, generator=g)
    dw_ref, db_ref = torch.autograd.grad(y, [w, b], bf16r(dy).double())
    To = y.shape[2]
    ws = torch.full((lib.avc_conv1d_wgrad_ws_floats(B, Cin, Cout, To, KS),), float("nan"), device=dev)
    dW = torch.full((Cout, Cin, KS), float("nan"), device=dev)
    db = torch.full((Cout,), float("nan"), device=dev)
    xd, dyd = to_pairs(x).to(dev), to_pairs(dy).to(dev)
    with op_dtype(lib, 3):
        rc = lib.avc_conv1d_wgrad(P(xd), xd.stride(0), xd.stride(1), 1, P(dyd), dyd.stride(0), dyd.stride(1), 1, 1, B, Cin, Cout, T, To, KS, stride,
                                  P(dW), P(db), P(ws), None)
    assert rc == 0, rc
    scale = dw_ref.abs().max().item()
    torch.testing.assert_close(dW.cpu(), dw_ref.float(), rtol=1e-4, atol=1e-5 * max(1.0, scale))     # fp32 accumulation of exact bf16 products
    torch.testing.assert_close(db.cpu(), db_ref.float(), rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------------------------
# the whole engine on pair storage (AVC_PLAN_BF16S; compute_dtype "bf16" / "bf16s")
# ---------------------------------------------------------------------------------------------------------------
def _tiny(act="relu"):
    return O.tiny_config(act=act)


def _flat_params(plan, sd, dev):
    from tests.test_engine import flat_params
    return flat_params(plan, sd, dev)


@pytest.mark.parametrize("kind", KINDS)
def test_storage_engine_shape_rules_and_fallback(kind):
    """The pair kernels take even channel counts and frame counts that are multiples of 4 at every level.  "bf16s" refuses other
    shapes; "bf16" falls back to the operand-rounding mode ("bf16r": fp32 storage) and says so."""
    from adaptive_voice_conversion_amd.engine import Plan
    lib, dev = backend(kind)
    cfg = _tiny()
    ok = Plan(cfg, 2, 32, lib=lib, compute_dtype="bf16")
    assert ok.compute_dtype == "bf16" and ok.pair_storage and lib.avc_plan_compute_dtype(ok.h) == 3
    from adaptive_voice_conversion_amd import engine as _engine
    _engine._warned.clear()
    with pytest.warns(RuntimeWarning, match="runs 'bf16r'"):       # (the fallback names itself and the offending shape, once)
        odd = Plan(cfg, 2, 34, lib=lib, compute_dtype="bf16")      # 34 -> 17 frames after the stride-2 block
    assert odd.compute_dtype == "bf16r" and not odd.pair_storage and lib.avc_plan_compute_dtype(odd.h) == 1
    with pytest.raises(RuntimeError, match="multiples of 4"):
        Plan(cfg, 2, 34, lib=lib, compute_dtype="bf16s")
    # half the activation bytes: the workspace of a pair plan is smaller than the fp32 plan's
    f32 = Plan(cfg, 2, 32, lib=lib)
    assert ok.workspace_floats < 0.8 * f32.workspace_floats
    # the compute dtype of a pair plan is a property of its workspace layout
    assert lib.avc_plan_set_compute_dtype(ok.h, 0) != 0


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("act", ["relu", "lrelu"])
def test_storage_engine_forward_inference_and_determinism(kind, act):
    """(1) forward of the train plan against the fp32 oracle (rel-L2 <= 3e-2, SURVEY 8c's bf16 bar), for both activations;
    (2) an inference plan (no gradient buffers) computes the same function bit for bit; (3) forward + backward twice: identical bits."""
    from adaptive_voice_conversion_amd.engine import Plan
    lib, dev = backend(kind)
    cfg = _tiny(act)
    B, T = 3, 32
    sd = O.make_state_dict(cfg, 2)
    x, eps = O.make_inputs(cfg, B, T, 2)
    xd, ed = x.to(dev), eps.to(dev)
    plan = Plan(cfg, B, T, lib=lib, compute_dtype="bf16s")
    params = _flat_params(plan, sd, dev)
    ws = torch.zeros(plan.workspace_floats, device=dev)
    plan.forward(params, xd, None, ed, ws)
    Cz = cfg["ContentEncoder"]["c_out"]
    shapes = {"muls": (B, 2 * Cz, plan.latent_len), "emb": (B, cfg["SpeakerEncoder"]["c_out"]), "dec": (B, cfg["Decoder"]["c_out"], plan.out_len)}
    got = {k: plan.view(ws, k, v).cpu().clone() for k, v in shapes.items()}
    mu, ls, emb, dec = O.ae_forward(x, eps, sd, cfg)
    ref = {"muls": torch.cat([mu, ls], 1), "emb": emb, "dec": dec}
    for k in got:
        e = ((got[k] - ref[k]).norm() / ref[k].norm()).item()
        assert e < 3e-2, (k, e)
    inf = Plan(cfg, B, T, lib=lib, compute_dtype="bf16s", mode="inference")
    assert inf.workspace_floats < plan.workspace_floats
    wsi = torch.zeros(inf.workspace_floats, device=dev)
    inf.forward(params, xd, None, ed, wsi)
    for k, v in shapes.items():
        assert torch.equal(inf.view(wsi, k, v).cpu(), got[k]), k
    plan.loss(xd, 10.0, ws)
    g1 = torch.zeros(plan.param_floats, device=dev)
    plan.backward(params, xd, None, ed, g1, ws, lambda_kl=1.0)
    g2 = torch.zeros_like(g1)
    plan.forward(params, xd, None, ed, ws)
    plan.loss(xd, 10.0, ws)
    plan.backward(params, xd, None, ed, g2, ws, lambda_kl=1.0)
    assert torch.isfinite(g1).all() and torch.equal(g1, g2), "the pair-storage backward is not deterministic"


@pytest.mark.gpu
@pytest.mark.parametrize("B,compute", [(64, "bf16s"), (256, "bf16s"), (256, "fp32")])
def test_backward_is_bit_reproducible_in_multi_stream_mode_at_bench_sizes(B, compute):
    """Round 6 regression.  The determinism check above runs a 3-sample tiny net -- one launch per stream at a time.  At the benchmarked sizes
    the backward pass runs the two encoder branches (and, from B = 128, the decoder's two half-batch chains) on two streams beside the
    weight-gradient streams, and the bf16 storage engine's gradients came out different from run to run (7 of 7 consecutive runs; 24 - 160 of
    the 166 tensors; forward outputs identical; the single-stream schedule reproducible): row kernels whose reductions the compiler had
    packed into v_pk_*_f32 instructions returned different sums beside the bf16 conv kernels of the other stream (scripts/pairs_race_probe.py,
    profiles/r06_pairs_race_*.txt).  The library is built without the packed-fp32 VALU forms since (csrc/build.sh); every run must now give
    the same bits -- on one workspace, and on a second one at another address."""
    from adaptive_voice_conversion_amd.engine import Plan
    lib, dev = backend("gpu")
    cfg = O.stock_config(80)
    sd = O.make_state_dict(cfg, 0)
    x, eps = O.make_inputs(cfg, B, 128, 0)
    plan = Plan(cfg, B, 128, lib=lib, compute_dtype=compute)
    params = _flat_params(plan, sd, dev)
    xd, ed = x.to(dev), eps.to(dev)

    def run(ws):
        plan.forward(params, xd, None, ed, ws)
        plan.loss(xd, 10.0, ws)
        g = torch.full((plan.param_floats,), float("nan"), device=dev)
        plan.backward(params, xd, None, ed, g, ws, lambda_kl=1.0)
        torch.cuda.synchronize()
        return g
    ws = torch.zeros(plan.workspace_floats, device=dev)
    ref = run(ws)
    assert torch.isfinite(ref).all()
    for i in range(6):
        assert torch.equal(run(ws), ref), f"run {i + 2} differs from run 1"
    ws2 = torch.full((plan.workspace_floats,), float("nan"), device=dev)
    assert torch.equal(run(ws2), ref), "a second workspace gives other bits"


@pytest.mark.parametrize("kind", KINDS)
def test_storage_engine_transposed_input_and_half_batch_chains(kind):
    """The [B, T, M] -> [B, M, T] view of data_utils.py:14-16 enters through the fp32 -> pair seam kernel (explicit strides), and the two
    half-batch decoder chains (dec_split_min) compute what the single chain computes."""
    from adaptive_voice_conversion_amd.engine import Plan
    lib, dev = backend(kind)
    cfg = _tiny()
    B, T = 4, 32
    sd = O.make_state_dict(cfg, 5)
    x, eps = O.make_inputs(cfg, B, T, 5)
    xt = x.transpose(1, 2).contiguous().to(dev)      # [B, T, M]
    xv = xt.transpose(1, 2)                           # the view the data pipeline hands over
    outs = []
    for split_min, xin in ((10 ** 6, x.to(dev)), (2, xv)):
        plan = Plan(cfg, B, T, lib=lib, compute_dtype="bf16s", tuning={"dec_split_min": split_min})
        params = _flat_params(plan, sd, dev)
        ws = torch.zeros(plan.workspace_floats, device=dev)
        plan.forward(params, xin, None, eps.to(dev), ws)
        plan.loss(xin, 10.0, ws)
        g = torch.zeros(plan.param_floats, device=dev)
        plan.backward(params, xin, None, eps.to(dev), g, ws, lambda_kl=1.0)
        outs.append((plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len)).cpu().clone(), g.cpu().clone()))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=0, atol=0)
    # (the half-batch split changes the split-K grouping of some weight-gradient launches: summation order only)
    assert ((outs[0][1] - outs[1][1]).norm() / outs[0][1].norm()).item() < 1e-4
