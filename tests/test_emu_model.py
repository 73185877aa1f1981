"""Host logic of the drop-in surface (AE / Solver / FusedClipAdam) exercised on the
CPU lane-level simulator: state_dict contract, autograd seam, one training step
vs the oracle, checkpoint round trip."""
import types

import pytest
import torch

from adaptive_voice_conversion_amd import _lib
from adaptive_voice_conversion_amd.model import AE
from adaptive_voice_conversion_amd.solver import Solver
from oracle import avc_oracle as O
from tests.emu_util import emu_lib


@pytest.fixture(scope="module")
def lib():
    return _lib.declare(emu_lib())


def test_state_dict_contract(lib):
    cfg = O.tiny_config()
    ae = AE(cfg, lib=lib)
    spec = O.param_spec(cfg)
    sd = ae.state_dict()
    assert [k for k, _ in spec] == list(sd.keys())
    assert all(tuple(sd[k].shape) == s for k, s in spec)
    ref = O.make_state_dict(cfg, 1)
    ae.load_state_dict(ref, strict=True)
    for (o, n, shape), (k, v) in zip(ae._layout, ref.items()):
        assert torch.equal(ae.flat_parameters()[o:o + n].view(shape), v), k  # parameters alias the flat buffer
    # stock config: 166 tensors, 4,892,880 (M=80) / 9,040,512 (M=512) elements
    n80 = sum(int(torch.tensor(s).prod()) for _, s in O.param_spec(O.stock_config(80)))
    n512 = sum(int(torch.tensor(s).prod()) for _, s in O.param_spec(O.stock_config(512)))
    assert (len(O.param_spec(O.stock_config(80))), n80, n512) == (166, 4892880, 9040512)


def test_unsupported_options_fail_loudly(lib):
    cfg = O.tiny_config()
    cfg["Decoder"]["sn"] = True
    with pytest.raises(NotImplementedError):
        AE(cfg, lib=lib)
    cfg = O.tiny_config()
    cfg["ContentEncoder"]["act"] = "lrelu"
    with pytest.raises(NotImplementedError):
        AE(cfg, lib=lib)


def test_autograd_seam_matches_oracle(lib):
    cfg = O.tiny_config()
    sd = O.make_state_dict(cfg, 4)
    x, eps = O.make_inputs(cfg, 2, 32, 4)
    ae = AE(cfg, lib=lib)
    ae.load_state_dict(sd)
    mu, ls, emb, dec = ae(x, eps=eps)
    loss_rec = torch.nn.L1Loss()(dec, x)
    loss_kl = 0.5 * torch.mean(torch.exp(ls) + mu ** 2 - 1 - ls)
    (10 * loss_rec + loss_kl).backward()
    outs, gref = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    torch.testing.assert_close(dec.detach(), outs["dec"], rtol=1e-4, atol=2e-5)
    for k, p in ae.named_parameters():
        d = gref[k].norm().item()
        e = (p.grad - gref[k]).norm().item()
        assert e <= 1e-4 * d + 1e-6, (k, e, d)


def test_solver_step_and_checkpoint_roundtrip(lib, tmp_path):
    cfg = O.tiny_config()
    sd = O.make_state_dict(cfg, 4)
    x, eps = O.make_inputs(cfg, 2, 32, 4)
    args = types.SimpleNamespace(store_model_path=str(tmp_path / "model"), load_model_path=str(tmp_path / "model"),
                                 load_model=False, data_dir=None, logdir=str(tmp_path / "log"), summary_steps=1,
                                 save_steps=1, tag="t")
    s = Solver(cfg, args, lib=lib)
    s.model.load_state_dict(sd)
    osd = {k: v.clone() for k, v in sd.items()}
    oopt = O.make_opt(osd, cfg)
    for it in range(2):
        meta = s.ae_step(x, 1.0, eps=eps)
        ometa, _, _ = O.ae_step(x, eps, osd, oopt, cfg, 1.0)
        tol = 1e-5 if it == 0 else 2e-3
        assert meta["loss_rec"] == pytest.approx(ometa["loss_rec"], rel=tol)
        assert meta["loss_kl"] == pytest.approx(ometa["loss_kl"], rel=tol)
        assert meta["grad_norm"] == pytest.approx(ometa["grad_norm"], rel=10 * tol)
        if it == 0:
            new = s.model.state_dict()
            bad = tot = 0
            for k in osd:
                diff = (new[k] - osd[k]).abs()
                bad += int((diff > 2e-6 + 1e-4 * osd[k].abs()).sum())
                tot += diff.numel()
                assert diff.max().item() <= 2.1 * cfg["optimizer"]["lr"]
            assert bad / tot < 5e-3  # only sign-flipped ~0-gradient elements may differ (see test_oracle_golden)
    assert s.kl_weight(0) == pytest.approx(1.0 / 20000) and s.kl_weight(30000) == 1.0
    s.save_model()
    s2 = Solver(cfg, args, lib=lib)
    s2.load_model()
    for (k, a), (_, b) in zip(s.model.state_dict().items(), s2.model.state_dict().items()):
        assert torch.equal(a, b), k
    assert s2.opt.step_count == 2 and torch.equal(s2.opt.vmax, s.opt.vmax)
    # the .opt file is loadable by torch.optim.Adam itself (reference solver.py:54)
    ref_params = [torch.nn.Parameter(v.clone()) for v in s.model.state_dict().values()]
    topt = torch.optim.Adam(ref_params, lr=1.0, amsgrad=True)
    topt.load_state_dict(torch.load(str(tmp_path / "model.opt")))
    assert topt.param_groups[0]["lr"] == cfg["optimizer"]["lr"]
