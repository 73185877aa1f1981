"""InstanceNorm / AdaIN / activation / residual rows inside the producing conv's epilogue (round 5, csrc/conv_shared.h: conv_epilogue_in)
against the oracle's restatement of the block halves (model.py:309-320 conv -> norm -> act [-> + pooled residual], :353-369 conv ->
[pixel shuffle] -> norm -> append_cond -> act [-> + upsampled residual]) -- and against the two-launch path (conv, then the row kernel):
same y bit for bit, same statistics and outputs to rounding (the sums run in another order)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from oracle import avc_oracle as O
from tests.emu_util import KINDS, P, backend
from tests.test_ops_conv import pack

GPU = pytest.mark.gpu


def run(lib, dev, x, w, b, stride, ops, cond, cond_off, relu, res, res_mode):
    B, Cin, Tin = x.shape
    Cout, _, KS = w.shape
    wp = pack(lib, dev, [w], 0)
    To = O.pad_conv(torch.zeros(1, Cin, Tin), torch.zeros(Cout, Cin, KS), None, stride).shape[2]
    C, T = Cout // ops, To * ops
    y = torch.full((B, C, T), float("nan"), device=dev)
    out = torch.full_like(y, float("nan"))
    mean = torch.full((B * C,), float("nan"), device=dev)
    rstd = torch.full_like(mean, float("nan"))
    fused = ctypes.c_int(-1)
    rc = lib.avc_conv1d_in_fwd(P(x), x.stride(0), x.stride(1), x.stride(2), B, Cin, Tin, P(wp), P(b), Cout, KS, stride, ops, P(y),
                               P(cond), cond.stride(0) if cond is not None else 0, cond_off, relu, P(res), res_mode,
                               res.shape[2] if res is not None else 0, P(out), P(mean), P(rstd), ctypes.byref(fused), None)
    assert rc == 0, rc
    return y, out, mean, rstd, fused.value


def reference(x, w, b, stride, ops, cond, cond_off, relu, res, res_mode):
    y = O.pad_conv(x, w, b, stride)
    if ops == 2:
        y = O.pixel_shuffle_1d(y, 2)
    C = y.shape[1]
    o = F.instance_norm(y, eps=1e-5)
    if cond is not None:
        beta, gamma = cond[:, cond_off:cond_off + C], cond[:, cond_off + C:cond_off + 2 * C]
        o = o * gamma[:, :, None] + beta[:, :, None]
    if relu == 1:
        o = torch.relu(o)
    elif relu == 2:
        o = F.leaky_relu(o, 0.01)
    if res is not None:
        o = o + {1: lambda r: r, 2: lambda r: O.avg_pool_ceil(r, 2), 5: lambda r: O.upsample_nearest(r, 2)}[res_mode](res)
    return y, o


# B, Cin, Cout, Tin, KS, stride, ops, affine, relu, res_mode, fused expected
CASES = [
    (3, 16, 32, 64, 5, 1, 1, False, 1, 0, 1),     # one sample per tile, plain IN + ReLU
    (5, 16, 40, 32, 5, 1, 1, False, 1, 1, 1),     # two samples per tile (odd batch: ragged last tile), identity residual, 40 of 64 rows valid
    (7, 24, 64, 16, 5, 1, 1, True, 2, 0, 1),      # four samples per tile, AdaIN + LeakyReLU
    (3, 16, 32, 64, 5, 2, 1, False, 1, 2, 1),     # stride 2 -> rows of 32, ceil-mode pooled residual (model.py:319)
    (3, 16, 64, 32, 5, 1, 2, True, 1, 5, 1),      # pixel shuffle: rows of 64 from conv rows (2c, 2c + 1), upsampled residual (model.py:362-369)
    (2, 16, 128, 64, 5, 1, 2, True, 1, 1, 1),     # ... rows of 128 frames out of 64-frame conv rows, two row tiles
    (9, 128, 128, 16, 5, 1, 1, False, 1, 0, 1),   # 16 K-chunks: the split-K wave groups in front of the fused epilogue
    (2, 16, 32, 128, 5, 1, 1, False, 1, 0, 0),    # rows of 128 frames: not fusable -> conv + row kernel
    (2, 16, 32, 24, 5, 1, 1, False, 1, 0, 0),     # a length that does not tile 64 columns: not fusable
    (3, 16, 32, 31, 5, 2, 1, False, 1, 2, 0),     # rows of 16 joined by the ceil-mode pool of a 31-frame residual (odd source, model.py:319):
                                                  # found by tests/test_engine_random_configs.py -- the row kernel keeps that case
    pytest.param(64, 128, 128, 64, 5, 1, 1, True, 1, 1, 1, marks=GPU),
    pytest.param(64, 128, 256, 32, 5, 1, 2, True, 1, 5, 1, marks=GPU),
    pytest.param(256, 128, 128, 32, 5, 2, 1, False, 1, 2, 1, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,Tin,KS,stride,ops,affine,relu,res_mode,want_fused", CASES)
def test_conv_in_fused_epilogue(kind, B, Cin, Cout, Tin, KS, stride, ops, affine, relu, res_mode, want_fused):
    if kind == "emu" and B * Cin * Cout * Tin * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 13 + Tin)
    x = torch.randn(B, Cin, Tin, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    C = Cout // ops
    To = O.pad_conv(torch.zeros(1, Cin, Tin), torch.zeros(Cout, Cin, KS), None, stride).shape[2] * ops
    cond = torch.randn(B, 4 * C + 8, generator=g) if affine else None
    cond_off = 2 * C + 8 if affine else 0
    Tres = {0: 0, 1: To, 2: (Tin if stride == 2 else 2 * To), 5: To // 2}[res_mode]   # (the pooled residual is the block's input: Tin frames)
    res = torch.randn(B, C, Tres, generator=g) if res_mode else None
    y_ref, o_ref = reference(x, w, b, stride, ops, cond, cond_off, relu, res, res_mode)
    d = lambda t: None if t is None else t.to(dev)
    y, out, mean, rstd, fused = run(lib, dev, d(x), d(w), d(b), stride, ops, d(cond), cond_off, relu, d(res), res_mode)
    assert fused == want_fused
    torch.testing.assert_close(y.cpu(), y_ref, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(out.cpu(), o_ref, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(mean.cpu().view(B, C), y_ref.mean(2), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rstd.cpu().view(B, C), 1.0 / torch.sqrt(y_ref.var(2, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-5)
    if fused:   # the two-launch path: the same conv kernel in front of the row kernel
        assert lib.avc_set_tuning(b"conv_in_fuse", 0) == 0
        try:
            y2, out2, mean2, rstd2, fused2 = run(lib, dev, d(x), d(w), d(b), stride, ops, d(cond), cond_off, relu, d(res), res_mode)
        finally:
            lib.avc_set_tuning(b"conv_in_fuse", 1)
        assert fused2 == 0
        assert torch.equal(y, y2)   # y = accumulator + bias either way
        torch.testing.assert_close(out, out2, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(mean, mean2, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rstd, rstd2, rtol=1e-5, atol=1e-6)


# ---- bf16 pair storage (compute_dtype "bf16"): the same fusion on pair tensors; y is rounded to bf16 first, statistics / output from the
# rounded values -- the two-launch path's arithmetic (conv -> pair rows -> instnorm_fwd_pairs_kernel), summation order aside
PAIR_CASES = [
    (3, 32, 32, 64, 5, 1, False, 1, 0),
    (5, 32, 40, 32, 5, 1, True, 1, 1),     # ragged last tile, 40 of 64 rows valid, AdaIN, identity residual
    (6, 48, 64, 16, 5, 1, True, 2, 0),     # four samples per tile, LeakyReLU
    (3, 32, 32, 64, 5, 2, False, 1, 2),    # stride 2 -> rows of 32 with the ceil-mode pooled residual
    (2, 32, 64, 32, 1, 1, True, 1, 5),     # 1x1 conv, upsampled residual
    pytest.param(64, 128, 128, 64, 5, 1, True, 1, 1, marks=GPU),
    pytest.param(256, 128, 128, 32, 5, 2, False, 1, 2, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,Tin,KS,stride,affine,relu,res_mode", PAIR_CASES)
def test_conv_in_fused_epilogue_on_pair_tensors(kind, B, Cin, Cout, Tin, KS, stride, affine, relu, res_mode):
    from tests.test_bf16_pairs import bf16r, close_bf16, from_pairs, op_dtype, to_pairs
    if kind == "emu" and B * Cin * Cout * Tin * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 17 + Tin)
    x = torch.randn(B, Cin, Tin, generator=g)
    w = torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5
    b = torch.randn(Cout, generator=g)
    To = O.pad_conv(torch.zeros(1, Cin, Tin), torch.zeros(Cout, Cin, KS), None, stride).shape[2]
    cond = torch.randn(B, 2 * Cout + 4, generator=g) if affine else None
    cond_off = 4 if affine else 0
    Tres = {0: 0, 1: To, 2: 2 * To, 5: To // 2}[res_mode]
    res = torch.randn(B, Cout, Tres, generator=g) if res_mode else None
    # reference: bf16 operands, y rounded once, everything after it from the rounded y, the output rounded once
    y_ref = bf16r(O.pad_conv(bf16r(x).double(), bf16r(w).double(), b.double(), stride).float())
    _, o_ref = reference_from_y(y_ref, cond, cond_off, relu, bf16r(res) if res is not None else None, res_mode)

    def go():
        xp, rp = to_pairs(x).to(dev), (to_pairs(res).to(dev) if res is not None else None)
        wd, bd, cd = w.to(dev), b.to(dev), (cond.to(dev) if cond is not None else None)
        wp = pack(lib, dev, [wd], 0)
        y = torch.zeros(B, Cout // 2, To, dtype=torch.int32, device=dev)
        out = torch.zeros_like(y)
        mean = torch.full((B * Cout,), float("nan"), device=dev)
        rstd = torch.full_like(mean, float("nan"))
        fused = ctypes.c_int(-1)
        rc = lib.avc_conv1d_in_fwd(P(xp), xp.stride(0), xp.stride(1), 1, B, Cin, Tin, P(wp), P(bd), Cout, KS, stride, 1, P(y), P(cd),
                                   cd.stride(0) if cd is not None else 0, cond_off, relu, P(rp), res_mode, Tres, P(out), P(mean), P(rstd),
                                   ctypes.byref(fused), None)
        assert rc == 0, rc
        return from_pairs(y.cpu()), from_pairs(out.cpu()), mean.cpu(), rstd.cpu(), fused.value

    with op_dtype(lib, 3):
        y, out, mean, rstd, fused = go()
        assert fused == 1
        assert lib.avc_set_tuning(b"conv_in_fuse", 0) == 0
        try:
            y2, out2, mean2, rstd2, fused2 = go()
        finally:
            lib.avc_set_tuning(b"conv_in_fuse", 1)
        assert fused2 == 0
    close_bf16(y, y_ref, ulps=2.0, atol=1e-3)      # (one bf16 ulp is up to 2^-7 of the value: an fp32 sum of 640 products lands on the other side of a rounding boundary now and then)
    close_bf16(out, o_ref, ulps=2.0, atol=2e-2)
    assert torch.equal(y, y2)                      # the same rounded conv output either way
    torch.testing.assert_close(mean, mean2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd, rstd2, rtol=1e-5, atol=1e-6)
    close_bf16(out, out2, ulps=2.0, atol=1e-6)     # (a statistic that differs in its last bit may move an output across a rounding boundary: one bf16 ulp, up to 2^-7 of the value)


def reference_from_y(y, cond, cond_off, relu, res, res_mode):
    C = y.shape[1]
    o = F.instance_norm(y, eps=1e-5)
    if cond is not None:
        beta, gamma = cond[:, cond_off:cond_off + C], cond[:, cond_off + C:cond_off + 2 * C]
        o = o * gamma[:, :, None] + beta[:, :, None]
    if relu == 1:
        o = torch.relu(o)
    elif relu == 2:
        o = F.leaky_relu(o, 0.01)
    if res is not None:
        o = o + {1: lambda r: r, 2: lambda r: O.avg_pool_ceil(r, 2), 5: lambda r: O.upsample_nearest(r, 2)}[res_mode](res)
    return y, o


# ---- the backward twin: InstanceNorm / AdaIN / activation BACKWARD inside the epilogue of the input-gradient launch that produces its g
# (autograd of model.py:309-320 / :353-369 one layer up).  Against the two-launch path (dgrad, then the row kernel): g bit for bit, dy / dcond
# to summation order -- the two-launch path itself is pinned to torch autograd by tests/test_ops_conv.py and tests/test_ops_rowops.py.
# B, Cin (= channels of the normalised rows), Cout, T (rows), KS, stride of the conv whose dgrad this is, res_mode (0, 1 identity, 3 pool^T, 4 up^T), affine, relu, fused expected
BWD_CASES = [
    (3, 32, 16, 64, 5, 1, 0, False, 1, 1),
    (5, 40, 16, 32, 5, 1, 1, True, 1, 1),      # two samples per tile, ragged last tile, 40 of 64 rows, AdaIN gradients, identity join
    (6, 64, 24, 16, 5, 1, 4, True, 2, 1),      # four samples per tile, the upsample's adjoint (model.py:61-63), LeakyReLU
    (3, 32, 16, 64, 5, 2, 3, False, 1, 1),     # stride-2 conv: its input gradient deals columns to the waves by parity; pool^T join
    (4, 32, 16, 32, 5, 2, 0, False, 1, 1),     # ... short rows
    (9, 128, 128, 16, 5, 1, 1, False, 1, 1),   # split-K wave groups in front of the fused epilogue
    (2, 32, 48, 32, 1, 1, 0, True, 1, 1),      # 1x1 (the heads / in_conv input gradients)
    (2, 32, 16, 128, 5, 1, 0, False, 1, 0),    # rows of 128 frames: not fusable
    pytest.param(64, 128, 128, 64, 5, 1, 1, True, 1, 1, marks=GPU),
    pytest.param(256, 128, 128, 32, 5, 2, 3, False, 1, 1, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,res_mode,affine,relu,want_fused", BWD_CASES)
def test_conv_dgrad_in_bwd_fused_epilogue(kind, B, Cin, Cout, T, KS, stride, res_mode, affine, relu, want_fused):
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 7 + T)
    Tdy = O.pad_conv(torch.zeros(1, Cin, T), torch.zeros(Cout, Cin, KS), None, stride).shape[2]
    w = (torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5).to(dev)
    dy = torch.randn(B, Cout, Tdy, generator=g).to(dev)
    y = torch.randn(B, Cin, T, generator=g)
    mean = y.mean(2).reshape(-1).to(dev)
    rstd = (1.0 / torch.sqrt(y.var(2, unbiased=False) + 1e-5)).reshape(-1).to(dev)
    y = y.to(dev)
    cond = torch.randn(B, 2 * Cin + 6, generator=g).to(dev) if affine else None
    coff = 6 if affine else 0
    Tres = {0: 0, 1: T, 3: T // 2, 4: 2 * T}[res_mode]
    res = torch.randn(B, Cin, Tres, generator=g).to(dev) if res_mode else None
    wpd = pack(lib, dev, [w], 1)

    def go():
        gout = torch.full((B, Cin, T), float("nan"), device=dev)
        dyo = torch.full((B, Cin, T), float("nan"), device=dev)
        dcond = torch.zeros_like(cond) if affine else None
        fused = ctypes.c_int(-1)
        rc = lib.avc_conv1d_dgrad_in_bwd(P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cout, Tdy, P(wpd), Cin, KS, stride, T, P(gout), P(res),
                                         res_mode, Tres, P(y), P(mean), P(rstd), P(cond), cond.stride(0) if affine else 0, coff, relu, P(dyo),
                                         P(dcond), dcond.stride(0) if affine else 0, coff, ctypes.byref(fused), None)
        assert rc == 0, rc
        return gout, dyo, dcond, fused.value

    g1, d1, c1, fused = go()
    assert fused == want_fused
    assert lib.avc_set_tuning(b"conv_in_fuse", 0) == 0
    try:
        g2, d2, c2, fused2 = go()
    finally:
        lib.avc_set_tuning(b"conv_in_fuse", 1)
    assert fused2 == 0
    assert torch.isfinite(d1).all() and torch.isfinite(g1).all()
    assert torch.equal(g1, g2)                                  # the same accumulators + the same join
    scale = d2.abs().max().item()
    torch.testing.assert_close(d1, d2, rtol=1e-4, atol=1e-5 * max(scale, 1.0))
    if affine:
        torch.testing.assert_close(c1, c2, rtol=1e-4, atol=1e-4)


# ... on bf16 pair tensors (round 6): g is rounded to bf16 first (what the two-launch path stores and the pair row kernel reads back), the
# InstanceNorm backward taken from the rounded values; g bit for bit, dy within a bf16 ulp of the two-launch path (a row sum that differs in its
# last bit may move an element across a rounding boundary), dcond to summation order.
PAIR_BWD_CASES = [
    (3, 32, 16, 64, 5, 1, 0, False, 1),
    (5, 40, 16, 32, 5, 1, 1, True, 1),       # two samples per tile, ragged last tile, 40 of 64 rows, AdaIN gradients, identity join
    (6, 64, 24, 16, 5, 1, 4, True, 2),       # four samples per tile, the upsample's adjoint, LeakyReLU
    (3, 32, 16, 64, 5, 2, 3, False, 1),      # stride-2 conv: columns dealt to the waves by parity; pool^T join
    (4, 32, 16, 32, 5, 2, 0, False, 1),
    (2, 32, 48, 32, 1, 1, 0, True, 1),       # 1x1
    pytest.param(64, 128, 128, 64, 5, 1, 1, True, 1, marks=GPU),
    pytest.param(256, 128, 128, 32, 5, 2, 3, False, 1, marks=GPU),
    pytest.param(128, 128, 256, 16, 5, 1, 4, True, 1, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,res_mode,affine,relu", PAIR_BWD_CASES)
def test_conv_dgrad_in_bwd_fused_epilogue_on_pair_tensors(kind, B, Cin, Cout, T, KS, stride, res_mode, affine, relu):
    from tests.test_bf16_pairs import bf16r, close_bf16, from_pairs, op_dtype, to_pairs
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 11 + T)
    Tdy = O.pad_conv(torch.zeros(1, Cin, T), torch.zeros(Cout, Cin, KS), None, stride).shape[2]
    w = (torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5).to(dev)
    dy = to_pairs(torch.randn(B, Cout, Tdy, generator=g)).to(dev)
    y = bf16r(torch.randn(B, Cin, T, generator=g))
    mean = y.mean(2).reshape(-1).to(dev)
    rstd = (1.0 / torch.sqrt(y.var(2, unbiased=False) + 1e-5)).reshape(-1).to(dev)
    yp = to_pairs(y).to(dev)
    cond = torch.randn(B, 2 * Cin + 6, generator=g).to(dev) if affine else None
    coff = 6 if affine else 0
    Tres = {0: 0, 1: T, 3: T // 2, 4: 2 * T}[res_mode]
    res = to_pairs(torch.randn(B, Cin, Tres, generator=g)).to(dev) if res_mode else None

    def go():
        wpd = pack(lib, dev, [w], 1)
        gout = torch.zeros(B, Cin // 2, T, dtype=torch.int32, device=dev)
        dyo = torch.zeros(B, Cin // 2, T, dtype=torch.int32, device=dev)
        dcond = torch.zeros_like(cond) if affine else None
        fused = ctypes.c_int(-1)
        rc = lib.avc_conv1d_dgrad_in_bwd(P(dy), dy.stride(0), dy.stride(1), 1, 1, B, Cout, Tdy, P(wpd), Cin, KS, stride, T, P(gout), P(res),
                                         res_mode, Tres, P(yp), P(mean), P(rstd), P(cond), cond.stride(0) if affine else 0, coff, relu, P(dyo),
                                         P(dcond), dcond.stride(0) if affine else 0, coff, ctypes.byref(fused), None)
        assert rc == 0, rc
        return from_pairs(gout.cpu()), from_pairs(dyo.cpu()), (dcond.cpu() if affine else None), fused.value

    with op_dtype(lib, 3):
        g1, d1, c1, fused = go()
        assert fused == 1
        assert lib.avc_set_tuning(b"conv_in_fuse", 0) == 0
        try:
            g2, d2, c2, fused2 = go()
        finally:
            lib.avc_set_tuning(b"conv_in_fuse", 1)
        assert fused2 == 0
    assert torch.isfinite(d1).all() and torch.isfinite(g1).all()
    assert torch.equal(g1, g2)                                  # the same accumulators + the same join, rounded once
    close_bf16(d1, d2, ulps=2.0, atol=1e-6 * max(d2.abs().max().item(), 1.0))
    if affine:
        torch.testing.assert_close(c1, c2, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kind", KINDS)
def test_fused_epilogues_keep_their_tile_under_the_wide_tile_switch(kind):
    """avc_tuning.tile12_wgs (opt-in: 64 x 128 tiles for k = 5 layers that keep the chip full) must not pull a launch that carries a fused
    InstanceNorm epilogue off its 64 x 64 tile: same result, still one launch."""
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 16, 64, generator=g).to(dev)
    w = (torch.randn(32, 16, 5, generator=g) / 9).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    ref = run(lib, dev, x, w, b, 1, 1, None, 0, 1, None, 0)
    assert lib.avc_set_tuning(b"tile12_wgs", 1) == 0
    try:
        got = run(lib, dev, x, w, b, 1, 1, None, 0, 1, None, 0)
    finally:
        lib.avc_set_tuning(b"tile12_wgs", 0)
    assert got[4] == 1 and ref[4] == 1
    for a, c in zip(ref[:4], got[:4]):
        assert torch.equal(a, c)
