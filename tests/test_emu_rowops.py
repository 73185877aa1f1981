"""InstanceNorm/AdaIN/ReLU/residual row kernels and the fused clip+Adam step on
the CPU lane-level simulator vs torch restatements (model.py:77-83, :296-369;
solver.py:75-77,:91-93)."""
import ctypes

import pytest
import torch

from oracle import avc_oracle as O
from tests.emu_util import I, L, P, emu_lib


def ref_block(y, cond, relu, res, res_mode):
    v = O.instance_norm(y)
    if cond is not None:
        v = O.append_cond(v, cond)
    if relu:
        v = torch.relu(v)
    if res is not None:
        if res_mode == 1:
            v = v + res
        elif res_mode == 2:
            v = v + O.avg_pool_ceil(res, 2)
        elif res_mode == 5:
            v = v + O.upsample_nearest(res, 2)
    return v


@pytest.mark.parametrize("B,C,T,affine,res_mode", [
    (3, 8, 16, True, 0), (2, 8, 32, False, 1), (2, 8, 64, True, 5), (2, 8, 128, True, 2), (1, 8, 24, True, 2),
    (1, 4, 1024, True, 1), (2, 8, 19, True, 2), (2, 8, 7, False, 5 - 5), (1, 4, 2048, False, 0), (3, 5, 12, True, 5),
])
def test_instnorm_fwd_bwd(B, C, T, affine, res_mode):
    lib = emu_lib()
    g = torch.Generator().manual_seed(T * 10 + B)
    y = (torch.randn(B, C, T, generator=g) * 2 + 0.5).requires_grad_(True)
    cond_all = torch.randn(B, 3 * 2 * C, generator=g)
    off = 2 * C
    cond = cond_all[:, off:off + 2 * C].clone().requires_grad_(True) if affine else None
    res = None
    Tres = 0
    if res_mode == 1:
        res = torch.randn(B, C, T, generator=g)
    elif res_mode == 2:
        Tres = 2 * T - (1 if T % 2 == 1 else 0)
        res = torch.randn(B, C, Tres, generator=g)
    elif res_mode == 5:
        if T % 2:
            pytest.skip("nearest x2 output is even")
        res = torch.randn(B, C, T // 2, generator=g)
    if res is not None:
        Tres = res.shape[2]
    ref = ref_block(y, cond, True, res, res_mode)
    out = torch.full((B, C, T), float("nan"))
    mean = torch.full((B * C,), float("nan"))
    rstd = torch.full((B * C,), float("nan"))
    rc = lib.avc_instnorm_fwd(P(y.detach()), B, C, T, P(cond_all if affine else None), L(cond_all.stride(0)), off, 1,
                              P(res), res_mode, Tres, P(out), P(mean), P(rstd), None)
    assert rc == 0
    torch.testing.assert_close(out, ref.detach(), rtol=1e-5, atol=1e-5)
    gout = torch.randn(B, C, T, generator=g)
    grads = torch.autograd.grad(ref, [y] + ([cond] if affine else []), gout)
    dy = torch.full((B, C, T), float("nan"))
    dcond = torch.zeros(B, 3 * 2 * C)
    rc = lib.avc_instnorm_bwd(P(gout), P(y.detach()), P(mean), P(rstd), B, C, T, P(cond_all if affine else None),
                              L(cond_all.stride(0)), off, 1, P(dy), P(dcond if affine else None), L(dcond.stride(0)),
                              off, None)
    assert rc == 0
    torch.testing.assert_close(dy, grads[0], rtol=2e-4, atol=2e-5)
    if affine:
        torch.testing.assert_close(dcond[:, off:off + 2 * C], grads[1], rtol=1e-4, atol=1e-4)
        assert dcond[:, :off].abs().max() == 0


@pytest.mark.parametrize("amsgrad,wd,prescale", [(True, 1e-4, 1.0), (False, 0.0, 0.5)])
def test_clip_adam_matches_torch(amsgrad, wd, prescale):
    lib = emu_lib()
    g = torch.Generator().manual_seed(3)
    n = 5000
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=5e-4, betas=(0.9, 0.999), amsgrad=amsgrad, weight_decay=wd)
    p = p0.clone()
    m = torch.zeros(n); v = torch.zeros(n); vmax = torch.zeros(n)
    lib.avc_clip_adam_ws_floats.restype = ctypes.c_long
    ws = torch.zeros(lib.avc_clip_adam_ws_floats(L(n)))
    gn = torch.zeros(1)
    F = ctypes.c_float
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (0.3 if step == 2 else 0.01)  # step 2 clips
        p_ref.grad = (grad * prescale).clone()
        gn_ref = torch.nn.utils.clip_grad_norm_([p_ref], 5.0)
        opt.step()
        gbuf = grad.clone()
        rc = lib.avc_clip_adam_step(P(p), P(gbuf), P(m), P(v), P(vmax), L(n), step, F(5e-4), F(0.9), F(0.999), F(1e-8),
                                    F(wd), int(amsgrad), F(5.0), F(prescale), 1, P(ws), P(gn), None)
        assert rc == 0
        assert gn.item() == pytest.approx(gn_ref.item(), rel=1e-5)
        torch.testing.assert_close(gbuf, p_ref.grad, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(p, p_ref.detach(), rtol=1e-5, atol=1e-6)
