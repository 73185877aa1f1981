"""Whole-model forward/backward of the engine on the CPU lane-level simulator
(tiny instance of the architecture) vs the oracle."""
import numpy as np
import pytest
import torch

from adaptive_voice_conversion_amd import _lib
from adaptive_voice_conversion_amd.engine import Plan
from oracle import avc_oracle as O
from tests.emu_util import emu_lib


def flat_params(plan, sd):
    flat = torch.zeros(plan.param_floats)
    for (off, n, shape), (k, v) in zip(plan.param_info, sd.items()):
        assert n == v.numel(), k
        flat[off:off + n] = v.reshape(-1)
    return flat


@pytest.fixture(scope="module")
def lib():
    return _lib.declare(emu_lib())


@pytest.mark.parametrize("B,T,transposed", [(2, 32, False), (3, 24, True)])
def test_engine_forward_backward_tiny(lib, B, T, transposed):
    cfg = O.tiny_config()
    sd = O.make_state_dict(cfg, 4)
    x, eps = O.make_inputs(cfg, B, T, 4)
    if transposed:
        x = x.transpose(1, 2).contiguous().transpose(1, 2)  # collate view, strides (T*M, 1, M)
    plan = Plan(cfg, B, T, lib=lib)
    assert plan.num_params == len(sd)
    for (off, n, shape), (k, v) in zip(plan.param_info, sd.items()):
        assert tuple(shape) == tuple(v.shape), k
    params = flat_params(plan, sd)
    ws = torch.full((plan.workspace_floats,), float("nan"))
    plan.forward(params, x, None, eps, ws)
    Tb = plan.latent_len
    Cz = cfg["ContentEncoder"]["c_out"]
    muls = plan.view(ws, "muls", (B, 2 * Cz, Tb))
    emb = plan.view(ws, "emb", (B, cfg["SpeakerEncoder"]["c_out"]))
    dec = plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len))
    outs, grads_ref = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    torch.testing.assert_close(emb, outs["emb"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(muls[:, :Cz], outs["mu"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(muls[:, Cz:], outs["log_sigma"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dec, outs["dec"], rtol=1e-4, atol=2e-5)

    plan.loss(x, cfg["lambda"]["lambda_rec"], ws)
    losses = plan.view(ws, "losses", (2,))
    assert losses[0].item() == pytest.approx(outs["loss_rec"].item(), rel=1e-5)
    assert losses[1].item() == pytest.approx(outs["loss_kl"].item(), rel=1e-5)
    grads = torch.full((plan.param_floats,), float("nan"))
    plan.backward(params, x, None, eps, grads, ws, lambda_kl=1.0)
    worst = 0.0
    for (off, n, shape), (k, gref) in zip(plan.param_info, grads_ref.items()):
        g = grads[off:off + n].view(shape)
        assert torch.isfinite(g).all(), k
        denom = gref.norm().item()
        err = (g - gref).norm().item()
        if denom > 1e-6:
            assert err / denom < 1e-4, (k, err / denom)
            worst = max(worst, err / denom)
        else:  # analytically-zero bias gradients
            assert err < 1e-6, (k, err)
    print("worst rel-L2 grad err", worst)


def test_engine_inference_odd_lengths(lib):
    cfg = O.tiny_config()
    sd = O.make_state_dict(cfg, 7)
    x, _ = O.make_inputs(cfg, 1, 37, 7)
    xc, _ = O.make_inputs(cfg, 1, 19, 14)
    plan = Plan(cfg, 1, 37, 19, lib=lib)
    params = flat_params(plan, sd)
    ws = torch.full((plan.workspace_floats,), float("nan"))
    plan.forward(params, x, xc, None, ws)
    dec = plan.view(ws, "dec", (1, cfg["Decoder"]["c_out"], plan.out_len))
    ref = O.ae_inference(x, xc, sd, cfg)
    assert dec.shape == ref.shape
    torch.testing.assert_close(dec, ref, rtol=1e-4, atol=2e-5)


def test_short_input_rejected_like_reference(lib):
    cfg = O.stock_config(80)
    with pytest.raises(RuntimeError, match="Padding size should be less"):
        Plan(cfg, 1, 16, lib=lib)
