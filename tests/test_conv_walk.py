"""The tile walk of conv_gemm (round 5: persistent workgroups over the column tiles of a row slab, the next tile's first chunk in flight
under the epilogue of this one) must give BIT-IDENTICAL results to the one-tile-per-workgroup launch: every tile is computed by the same
instruction sequence (model.py:21-32 semantics are covered by tests/test_ops_conv.py; this file only compares the two schedules).
avc_set_tuning("conv_walk", -W) forces W walkers per row slab whatever the chip would hold.  WALK instances exist for the exact-fp32 k = 5
chunk (forward and mirrored input gradient), the grouped bank launch and the 1x1 convs; the other cases of the table check that a launch
without such an instance quietly keeps one tile per workgroup."""
import pytest
import torch

from oracle import avc_oracle as O
from tests.emu_util import KINDS, P, backend
from tests.test_ops_conv import conv_fwd, pack

GPU = pytest.mark.gpu


def _tuned(lib, **kw):
    for k, v in kw.items():
        assert lib.avc_set_tuning(k.encode(), v) == 0, k


def _dgrad(lib, dev, dy, w, T, stride, tile, res=None, res_mode=0, mask=None):
    B, Cout, To = dy.shape
    Cin, KS = w.shape[1], w.shape[2]
    wpd = pack(lib, dev, [w], 1)
    dx = torch.full((B, Cin, T), float("nan"), device=dev)
    dx2 = torch.full((B, Cin, T), float("nan"), device=dev) if mask is not None else None
    rb = rc = rt = Tres = 0
    if res is not None:
        rb, rc, rt, Tres = res.stride(0), res.stride(1), res.stride(2), res.shape[2]
    rcode = lib.avc_conv1d_dgrad(P(dy), dy.stride(0), dy.stride(1), dy.stride(2), 1, B, Cout, To, P(wpd), Cin, KS, stride, T, P(dx),
                                 dx.stride(0), dx.stride(1), dx.stride(2), P(res), res_mode, rb, rc, rt, Tres, P(dx2), P(mask), tile, None)
    assert rcode == 0, rcode
    return dx, dx2


# B, Cin, Cout, T, KS, stride, tile, walkers
CASES = [
    (6, 16, 32, 64, 5, 1, 11, 2),     # one tile per sample: 6 tiles on 2 / 4 walkers (uneven walk)
    (6, 16, 32, 64, 5, 1, 11, 4),
    (5, 16, 32, 130, 5, 1, 11, 3),    # three tiles per sample (ragged last): walkers keep their first frame
    (5, 16, 32, 130, 5, 1, 11, 6),
    (4, 24, 40, 128, 5, 2, 11, 2),    # stride 2 (dgrad: one column parity per wave)
    (12, 16, 32, 16, 5, 1, 11, 2),    # four short samples per tile, three tiles
    (12, 16, 32, 32, 5, 2, 11, 3),    # stride 2, short rows
    (5, 40, 32, 70, 1, 1, 11, 2),     # 1x1, 32-channel chunks, ragged second tile
    (4, 16, 130, 64, 3, 1, 21, 2),    # 128-row tile
    (3, 16, 64, 256, 5, 1, 12, 2),    # 64 x 128 tile
    (7, 20, 32, 64, 5, 1, 11, 3),     # reduction channels not a multiple of 8
    pytest.param(64, 128, 128, 128, 5, 1, 0, 32, marks=GPU),
    pytest.param(64, 128, 128, 128, 5, 2, 0, 16, marks=GPU),
    pytest.param(32, 1104, 128, 128, 1, 1, 0, 8, marks=GPU),
    pytest.param(64, 128, 256, 16, 5, 1, 11, 4, marks=GPU),
]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,Cin,Cout,T,KS,stride,tile,W", CASES)
def test_conv_walk_bit_identical(kind, B, Cin, Cout, T, KS, stride, tile, W):
    if kind == "emu" and B * Cin * Cout * T * KS > 3e7:
        pytest.skip("gpu-sized")
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(B * 31 + T)
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, KS, generator=g) / (Cin * KS) ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    To = O.pad_conv(torch.zeros(1, Cin, T), torch.zeros(Cout, Cin, KS), None, stride).shape[2]
    res = torch.randn(B, Cout, To, generator=g).to(dev)
    dy = torch.randn(B, Cout, To, generator=g).to(dev)
    gres = torch.randn(B, Cin, T, generator=g).to(dev)
    mask = torch.randn(B, Cin, T, generator=g).to(dev)
    outs = {}
    try:
        for name, walk in (("tile", 0), ("walk", -W)):
            _tuned(lib, kg_wgs=0, conv_walk=walk)   # (no split-K wave groups: they keep one tile per workgroup)
            y, y2 = conv_fwd(lib, dev, x, w, b, stride, act=1, tile=tile, res=res, res_mode=1)
            dx, dx2 = _dgrad(lib, dev, dy, w, T, stride, tile, res=gres, res_mode=1, mask=mask)
            outs[name] = (y, y2, dx, dx2)
    finally:
        _tuned(lib, kg_wgs=256, conv_walk=0)
    for a, c in zip(outs["tile"], outs["walk"]):
        assert torch.isfinite(a).all()
        assert torch.equal(a, c)
    # ... and the one-tile result is the function (spot check; the table-driven tests cover it)
    ref = torch.relu(O.pad_conv(x.cpu(), w.cpu(), b.cpu(), stride))
    torch.testing.assert_close(outs["walk"][0].cpu(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("kind", KINDS)
def test_conv_walk_refuses_a_ragged_last_group_of_short_samples(kind):
    """Short rows walk in whole groups of samples only (every tile of a walk must have the first tile's geometry): with B % samples-per-tile
    != 0 the launcher silently keeps one tile per workgroup -- same result either way."""
    lib, dev = backend(kind)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(11, 16, 16, generator=g).to(dev)
    w = (torch.randn(32, 16, 5, generator=g) / 9).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    try:
        _tuned(lib, kg_wgs=0, conv_walk=0)
        y0, _ = conv_fwd(lib, dev, x, w, b, 1, act=1, tile=11)
        _tuned(lib, kg_wgs=0, conv_walk=-2)
        y1, _ = conv_fwd(lib, dev, x, w, b, 1, act=1, tile=11)
    finally:
        _tuned(lib, kg_wgs=256, conv_walk=0)
    assert torch.equal(y0, y1)
