"""Whole-model parity: engine forward / loss / backward / train step vs the oracle
and vs the committed reference goldens.

kind='emu': CPU lane-level simulation of the same kernels on a tiny instance of
the architecture.  kind='gpu': the gfx950 library through the C ABI on the stock
80-mel / 512-mel configs.  Tolerances (fp32): forward atol 2e-5 /
rtol 1e-4; per-tensor gradient rel-L2 <= 1e-4 (abs <= 1e-6 on the 23
analytically-zero bias gradients)."""
import os

import numpy as np
import pytest
import torch

from adaptive_voice_conversion_amd.engine import Plan
from oracle import avc_oracle as O
from tests.emu_util import KINDS, backend

GPU = pytest.mark.gpu


def flat_params(plan, sd, dev):
    flat = torch.zeros(plan.param_floats)
    for (off, n, shape), (k, v) in zip(plan.param_info, sd.items()):
        assert n == v.numel() and tuple(shape) == tuple(v.shape), k
        flat[off:off + n] = v.reshape(-1)
    return flat.to(dev)


def zero_grad_bias(name, cfg):
    """Biases of convs that feed an affine-free InstanceNorm directly have an analytically zero
    gradient: every content-encoder conv, the decoder's in_conv / first convs and the
    second convs that are not pixel-shuffled."""
    if not name.endswith(".bias"):
        return False
    if name.startswith("content_encoder.") and (".in_conv_layer." in name or "_conv_layers." in name):
        return True
    if name.startswith("decoder."):
        if ".in_conv_layer." in name or ".first_conv_layers." in name:
            return True
        if ".second_conv_layers." in name:
            return cfg["Decoder"]["upsample"][int(name.split(".")[2])] == 1
    return False


def check_grads(plan, grads, grads_ref, tol=1e-4, cfg=None, zero_abs=1e-6):
    """Per-tensor gradient parity: rel-L2 <= 1e-4, abs <= 1e-6 on the analytically-zero bias
    gradients (BASELINE.md tolerance).  Returns (worst tensor, median tensor, whole gradient)."""
    g = grads.cpu()
    errs, worst, num, den = [], 0.0, 0.0, 0.0
    for (off, n, shape), (k, gref) in zip(plan.param_info, grads_ref.items()):
        gi = g[off:off + n].view(shape)
        assert torch.isfinite(gi).all(), k
        denom, err = gref.norm().item(), (gi - gref).norm().item()
        num += err ** 2
        den += denom ** 2
        if not zero_grad_bias(k, cfg):
            errs.append(err / max(denom, 1e-30))
            if tol is not None:
                assert err / denom < tol, (k, err / denom)
            worst = max(worst, err / denom)
        else:  # analytically-zero bias gradients: pure fp32 noise on both sides
            assert denom < 1e-4 and err < zero_abs, (k, err, denom)
    return worst, sorted(errs)[len(errs) // 2], (num / den) ** 0.5


def branch_matched_oracle(plan, ws, x, eps, sd, cfg):
    """Oracle losses/gradients evaluated on the piecewise-linear branch the ENGINE took.

    A ReLU whose pre-activation is 0 +- fp32 roundoff is decided differently by any two correct
    fp32 implementations (density ~0.4/unit x ~1e-6 noise x 6e5 ReLU sites per segment = about
    one flip per pair of segments; the reference's own fp32 and fp64 runs disagree the same way,
    measured: 2.9e-3 on speaker_encoder.conv_bank.2.weight at B=2).  One flipped element moves the
    small tensor it feeds by ~1e-2 and everything upstream by ~1e-3, which says nothing about
    kernel correctness.  So the strict 1e-4 comparison is made with the oracle's ReLUs driven by the
    engine's masks (avc_plan_relu_site, recomputed exactly from the engine's saved tensors): both
    sides then differentiate the same function.  Forward outputs are compared WITHOUT this aid."""
    masks = [m.cpu() for m in plan.relu_masks(ws)]
    with O.relu_masks(masks):
        return O.loss_and_grads(x, eps, sd, cfg, 1.0)


CASES = [
    ("emu", "tiny", 2, 32, False), ("emu", "tiny", 3, 24, True),
    ("emu", "tiny_lrelu", 3, 40, False),   # act: lrelu in all three networks
    ("emu", "tiny", 19, 16, False),        # two workgroups of the 16-sample dense-stack kernel, the second one partial
    pytest.param("gpu", "tiny_lrelu", 5, 64, False, marks=GPU),
    ("emu", "tiny128", 2, 40, False),   # 128 hidden channels
    ("emu", "tiny128x3", 2, 40, False), # ... + tuning conv_x3 = 2: the split-bf16 conv kernel in every eligible layer of the plan
    pytest.param("gpu", "m80x3", 8, 128, False, marks=GPU),   # split-bf16 kernel, every eligible layer
    pytest.param("gpu", "m80x3", 3, 40, False, marks=GPU),    # ... odd lengths
    pytest.param("gpu", "tiny", 3, 24, True, marks=GPU),
    pytest.param("gpu", "m80", 4, 128, True, marks=GPU),
    pytest.param("gpu", "m80", 2, 256, False, marks=GPU),
    ("emu", "tiny8", 2, 32, False),                        # 8 blocks per encoder: the deepest plan
    ("emu", "tiny8b", 2, 32, False),                       # ... with wgrad_batch = 64: every branch's weight gradients in ONE flush (up to 16 layers in a stream-K launch)
    ("emu", "tiny_nd0", 2, 32, False),                     # n_dense_blocks = 0: the dense stack is the output layer alone (crashed the backward until round 4)
    pytest.param("gpu", "tiny_nd0", 20, 32, False, marks=GPU),
    ("emu", "tiny_early", 3, 32, False),                   # decoder weight gradients flushed under the decoder's own backward chain (dec_wgrad_flush)
    pytest.param("gpu", "m80_early", 4, 128, False, marks=GPU),
    pytest.param("gpu", "tiny8", 3, 64, False, marks=GPU),
    pytest.param("gpu", "tiny8b", 3, 64, False, marks=GPU),
    pytest.param("gpu", "m80", 3, 24, False, marks=GPU),   # T_l = 3 at the bottleneck
    pytest.param("gpu", "m80", 3, 40, False, marks=GPU),   # T_l = 5: odd rows, generic InstanceNorm path
    pytest.param("gpu", "m512", 2, 128, False, marks=GPU),
]


def deep_tiny_config():
    """The deepest network a plan takes (8 blocks per encoder = AVC_MAX_BLOCKS, 6 in the decoder: its 12 AdaIN affines are one stacked
    operand): 16 k = 5 weight gradients per encoder branch and 12 held in the decoder -- with `wgrad_batch = 64` ("tiny8b") a branch's flush
    is as many layers as a stream-K launch has descriptors for (AVC_WGRAD_MAXL = 16)."""
    cfg = O.tiny_config(n_blocks=8)
    for k in ("SpeakerEncoder", "ContentEncoder"):
        cfg[k]["subsample"] = [1, 1, 1, 2, 1, 1, 1, 2]
    cfg["Decoder"]["n_conv_blocks"] = 6
    cfg["Decoder"]["upsample"] = [2, 1, 1, 2, 1, 1]
    return cfg


def get_cfg(name):
    return {"tiny_nd0": lambda: O.tiny_config(n_dense=0), "tiny8": deep_tiny_config, "tiny8b": deep_tiny_config, "tiny_early": O.tiny_config, "m80_early": lambda: O.stock_config(80), "tiny": O.tiny_config, "tiny_lrelu": lambda: O.tiny_config(act="lrelu"), "tiny128": lambda: O.tiny_config(n_mels=16, c_h=128, c_bank=32, bank_size=4, n_blocks=2, n_dense=1), "m80": lambda: O.stock_config(80), "m80x3": lambda: O.stock_config(80), "tiny128x3": lambda: O.tiny_config(n_mels=16, c_h=128, c_bank=32, bank_size=4, n_blocks=2, n_dense=1), "m512": lambda: O.stock_config(512)}[name]()


@pytest.mark.parametrize("kind,cfgname,B,T,transposed", CASES)
def test_forward_loss_backward_vs_oracle(kind, cfgname, B, T, transposed):
    lib, dev = backend(kind)
    cfg = get_cfg(cfgname)
    sd = O.make_state_dict(cfg, 4)
    x, eps = O.make_inputs(cfg, B, T, 4)
    xd = x.to(dev)
    if transposed:
        xd = xd.transpose(1, 2).contiguous().transpose(1, 2)  # collate view, strides (T*M, 1, M)
    x3 = cfgname.endswith("x3")            # opt-in split-bf16 products: conv kernel (2 = every layer of an eligible shape, whatever its size) + weight gradients
    plan = Plan(cfg, B, T, lib=lib, tuning={"conv_x3": 2, "wgrad_x3": 1} if x3 else ({"wgrad_batch": 64} if cfgname == "tiny8b" else ({"dec_wgrad_flush": 2, "dec_wgrad_wgs": 24} if cfgname.endswith("_early") else None)))
    assert plan.num_params == len(sd)
    if x3:   # the opt-in kernel brings its own weight images: the plan really switched
        assert plan.workspace_floats > Plan(cfg, B, T, lib=lib).workspace_floats
    params = flat_params(plan, sd, dev)
    ws = torch.full((plan.workspace_floats,), float("nan"), device=dev)
    plan.forward(params, xd, None, eps.to(dev), ws)
    Tb, Cz = plan.latent_len, cfg["ContentEncoder"]["c_out"]
    muls = plan.view(ws, "muls", (B, 2 * Cz, Tb)).cpu()
    emb = plan.view(ws, "emb", (B, cfg["SpeakerEncoder"]["c_out"])).cpu()
    dec = plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len)).cpu()
    outs, grads_ref = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    torch.testing.assert_close(emb, outs["emb"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(muls[:, :Cz], outs["mu"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(muls[:, Cz:], outs["log_sigma"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dec, outs["dec"], rtol=1e-4, atol=2e-5)
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    losses = plan.view(ws, "losses", (2,)).cpu()
    assert losses[0].item() == pytest.approx(outs["loss_rec"].item(), rel=1e-5)
    assert losses[1].item() == pytest.approx(outs["loss_kl"].item(), rel=1e-5)
    grads = torch.full((plan.param_floats,), float("nan"), device=dev)
    plan.backward(params, xd, None, eps.to(dev), grads, ws, lambda_kl=1.0)
    outs_m, grads_m = branch_matched_oracle(plan, ws, x, eps, sd, cfg)
    torch.testing.assert_close(outs_m["dec"], outs["dec"], rtol=1e-4, atol=2e-5)   # same function value either way
    # T=24 reaches 3-frame rows at the bottleneck: InstanceNorm over 3 samples is ill-conditioned in
    # fp32 (the oracle's own fp32 vs fp64 gradients differ by 2.1e-4 on decoder.in_conv_layer.weight
    # there, 6e-6 at T=48; measured), so that case gets 5e-3; everything else the stated 1e-4.
    illc = T <= 24 and not cfgname.startswith("tiny")
    # (the 23 analytically-zero bias gradients are sums of rounding errors: 1e-6 for the stock depth; the 8-block net measures 1.01e-6)
    zabs = 2e-5 if illc else (3e-6 if cfgname.startswith("tiny8") else 1e-6)
    worst, med, total = check_grads(plan, grads, grads_m, tol=5e-3 if illc else 1e-4, cfg=cfg, zero_abs=zabs)
    assert med < 2e-5
    uw, um, ut = check_grads(plan, grads, grads_ref, tol=None, cfg=cfg, zero_abs=zabs)
    print(f"[{kind}/{cfgname} B={B} T={T}] grad rel-L2 (same ReLU branch): worst tensor {worst:.2e}, median {med:.2e}, "
          f"whole gradient {total:.2e} | vs the oracle's own branch: worst {uw:.2e}, median {um:.2e}, whole {ut:.2e}")
    assert ut < 3e-2  # even with kink flips the whole gradient stays close


@pytest.mark.parametrize("kind,cfgname,Ts,Tc", [("emu", "tiny", 37, 19), pytest.param("gpu", "m80", 100, 77, marks=GPU),
                                                pytest.param("gpu", "m80", 333, 129, marks=GPU)])
def test_inference_odd_unequal_lengths(kind, cfgname, Ts, Tc):
    lib, dev = backend(kind)
    cfg = get_cfg(cfgname)
    sd = O.make_state_dict(cfg, 7)
    x, _ = O.make_inputs(cfg, 1, Ts, 7)
    xc, _ = O.make_inputs(cfg, 1, Tc, 14)
    plan = Plan(cfg, 1, Ts, Tc, lib=lib)
    params = flat_params(plan, sd, dev)
    ws = torch.full((plan.workspace_floats,), float("nan"), device=dev)
    plan.forward(params, x.to(dev), xc.to(dev), None, ws)
    dec = plan.view(ws, "dec", (1, cfg["Decoder"]["c_out"], plan.out_len)).cpu()
    ref = O.ae_inference(x, xc, sd, cfg)
    assert dec.shape == ref.shape  # T' = 8*ceil(T/8)
    torch.testing.assert_close(dec, ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("kind", KINDS)
def test_short_input_rejected_like_reference(kind):
    lib, _ = backend(kind)
    with pytest.raises(RuntimeError, match="Padding size should be less"):
        Plan(O.stock_config(80), 1, 16, lib=lib)


X3 = dict(compute_dtype="fp32x3", tuning={"conv_x3": 2})   # opt-in split-bf16 products in EVERY eligible conv layer + the weight gradients


@pytest.mark.gpu
@pytest.mark.parametrize("compute", ["fp32", "fp32x3"])
@pytest.mark.parametrize("name,cfgname", [("train_m80_t128_b2", "m80"), ("train_m80_t128_b4_s1", "m80"), ("train_m80_t64_b1_full", "m80"),
                                          ("train_m80_t256_b1", "m80"), ("train_m512_t128_b1", "m512"),
                                          ("train_tiny_t32_b2", "tiny"), ("train_tiny_t24_b3", "tiny")])
def test_gpu_matches_reference_goldens(name, cfgname, compute, golden_dir):
    """HIP path vs fixtures produced by the REAL reference (oracle/make_golden.py); the opt-in fp32x3 mode against the same bars."""
    from oracle.make_golden import tensor_stats
    lib, dev = backend("gpu")
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = get_cfg(cfgname)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    sd = O.make_state_dict(cfg, seed)
    x, eps = O.make_inputs(cfg, B, T, seed)
    plan = Plan(cfg, B, T, lib=lib, **(X3 if compute == "fp32x3" else {}))
    params = flat_params(plan, sd, dev)
    ws = torch.zeros(plan.workspace_floats, device=dev)
    xd, ed = x.to(dev), eps.to(dev)
    plan.forward(params, xd, None, ed, ws)
    Tb, Cz = plan.latent_len, cfg["ContentEncoder"]["c_out"]
    muls = plan.view(ws, "muls", (B, 2 * Cz, Tb)).cpu()
    emb = plan.view(ws, "emb", (B, cfg["SpeakerEncoder"]["c_out"])).cpu()
    dec = plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len)).cpu()
    mine = np.stack([tensor_stats(t) for t in (muls[:, :Cz], muls[:, Cz:], emb, dec)])
    np.testing.assert_allclose(mine, g["out_stats"], rtol=1e-4, atol=2e-5)
    if "dec" in g:
        np.testing.assert_allclose(dec.numpy(), g["dec"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(muls[:, :Cz].numpy(), g["mu"], rtol=1e-4, atol=2e-5)
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    losses = plan.view(ws, "losses", (2,)).cpu()
    assert losses[0].item() == pytest.approx(float(g["loss_rec_0"]), rel=2e-5)
    assert losses[1].item() == pytest.approx(float(g["loss_kl_0"]), rel=2e-5)
    grads = torch.zeros(plan.param_floats, device=dev)
    plan.backward(params, xd, None, ed, grads, ws, lambda_kl=1.0)
    gc = grads.cpu()
    gs = np.stack([tensor_stats(gc[o:o + n].view(shape)) for o, n, shape in plan.param_info])
    ref = g["grad_stats"]
    # The fixtures hold the reference's gradients on ITS ReLU branch; a kink flip (see
    # branch_matched_oracle) moves single tensors by up to ~1e-2, so the fixture comparison is
    # strict on the small instance and statistical on the stock configs; the strict per-tensor
    # 1e-4 pin on the stock configs is test_forward_loss_backward_vs_oracle (branch-matched).
    strict = cfgname == "tiny"
    np.testing.assert_allclose(gs[:, 0], ref[:, 0], rtol=1e-4 if strict else 3e-2, atol=1e-6)   # per-tensor L2 norms
    bad = np.abs(gs[:, 3:] - ref[:, 3:]) > (5e-6 + 5e-3 * np.abs(ref[:, 3:]))                   # sampled entries
    assert bad.mean() <= (0.0 if strict else 0.10), bad.mean()
    total = float(np.sqrt((gs[:, 0] ** 2).sum()))
    assert total == pytest.approx(float(g["grad_norm_0"]), rel=1e-4 if strict else 5e-3)


@pytest.mark.parametrize("kind,name,cfgname,compute", [("emu", "train_tiny_t32_b2", "tiny", "fp32"),
                                                       ("emu", "train_tiny_lrelu_t32_b2", "tiny_lrelu", "fp32"),   # act: lrelu (model.py:93-99), every tensor vs the REAL reference
                                                       pytest.param("gpu", "train_tiny_lrelu_t32_b2", "tiny_lrelu", "fp32", marks=GPU),
                                                       pytest.param("gpu", "train_tiny_t32_b2", "tiny", "fp32", marks=GPU),
                                                       pytest.param("gpu", "train_m80_t64_b1_full", "m80", "fp32", marks=GPU),
                                                       pytest.param("gpu", "train_m80_t128_b2", "m80", "fp32", marks=GPU),
                                                       pytest.param("gpu", "train_m80_t64_b1_full", "m80", "fp32x3", marks=GPU),
                                                       pytest.param("gpu", "train_m80_t128_b2", "m80", "fp32x3", marks=GPU)])
def test_complete_gradient_tensors_vs_reference_golden(kind, name, cfgname, compute, golden_dir):
    """Strict pin that takes NOTHING from the engine to drive a checker: the fixture holds complete gradient tensors
    produced by the REAL reference (every bias + one block of each network; every tensor of the tiny net) and the
    reference's own 0/1 ReLU decisions.  The margin fixtures were generated from seeds whose forward pass keeps every
    pre-activation >= 4e-6 (stock 80-mel, 3e5 sites) / 2e-5 (tiny) away from a kink, so the engine must take exactly the
    reference's branch there and every stored tensor must agree to rel-L2 1e-4.  The B=2, T=128 fixture has sites ~1e-7
    from a kink (best of 400 seeds): the engine may differ from the reference only at recorded near-kink sites, and the
    1e-4 bar applies when no site differs (else the kink-flip bar of test_gpu_matches_reference_goldens, 3e-2)."""
    lib, dev = backend(kind)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = get_cfg(cfgname)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    sd = O.make_state_dict(cfg, seed)
    x, eps = O.make_inputs(cfg, B, T, seed)
    plan = Plan(cfg, B, T, lib=lib, **(X3 if compute == "fp32x3" else {}))
    params = flat_params(plan, sd, dev)
    ws = torch.zeros(plan.workspace_floats, device=dev)
    xd, ed = x.to(dev), eps.to(dev)
    plan.forward(params, xd, None, ed, ws)
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    grads = torch.zeros(plan.param_floats, device=dev)
    plan.backward(params, xd, None, ed, grads, ws, lambda_kl=1.0)
    masks = plan.relu_masks(ws)
    assert [m.numel() for m in masks] == list(g["relu_sizes"])
    n = int(g["relu_sizes"].sum())
    ref_bits = np.unpackbits(g["relu_bits"])[:n].astype(bool)
    mine = np.concatenate([m.reshape(-1).cpu().numpy() for m in masks])
    diff = np.nonzero(mine != ref_bits)[0]
    assert np.isin(diff, g["relu_near_idx"]).all(), "a ReLU decision differs from the reference's away from any kink"
    margin_fixture = float(g["relu_margin"]) >= 4e-6
    if margin_fixture:
        assert diff.size == 0, (diff[:10], g["relu_margin"])
    tol = 1e-4 if diff.size == 0 else 3e-2
    gc, worst, checked = grads.cpu(), 0.0, 0
    by_name = {k: (off, nn_, shape) for (off, nn_, shape), k in zip(plan.param_info, sd)}
    for k in g.files:
        if not k.startswith("gradfull/"):
            continue
        off, nn_, shape = by_name[k[9:]]
        gi, ref = gc[off:off + nn_].view(shape), torch.from_numpy(g[k])
        d = ref.norm().item()
        if zero_grad_bias(k[9:], cfg):
            assert d < 1e-4 and (gi - ref).abs().max().item() < 2e-6, k
        else:
            e = (gi - ref).norm().item() / d
            worst = max(worst, e)
            assert e < tol, (k, e, diff.size)
        checked += 1
    assert checked >= 60
    print(f"[{kind}/{name}/{compute}] {checked} complete gradient tensors vs the REAL reference, no branch matching: worst rel-L2 {worst:.2e} "
          f"(bar {tol:g}); ReLU decisions differing from the reference's: {diff.size} of {n} (closest recorded site {float(g['relu_margin']):.1e})")


@pytest.mark.gpu
def test_gpu_full_size_properties():
    """BASELINE config 2 size (B=256, 80x128): properties that need no oracle run.
    (1) data-parallel exactness: every sample's outputs equal those of the same
    sample in a smaller batch (InstanceNorm has no batch coupling);
    (2) run-to-run determinism (split-K slabs are reduced in a fixed order);
    (3) gradient of a batch = sum of gradients of its two halves (linearity of the
    wgrad reduction), with mean-loss scaling."""
    lib, dev = backend("gpu")
    cfg = O.stock_config(80)
    sd = O.make_state_dict(cfg, 0)
    B, T = 256, 128
    x, eps = O.make_inputs(cfg, B, T, 0)
    xd, ed = x.to(dev), eps.to(dev)
    big = Plan(cfg, B, T, lib=lib)
    params = flat_params(big, sd, dev)
    ws = torch.zeros(big.workspace_floats, device=dev)
    big.forward(params, xd, None, ed, ws)
    dec = big.view(ws, "dec", (B, 80, T)).clone()
    emb = big.view(ws, "emb", (B, 128)).clone()
    big.loss(xd, 10.0, ws)
    g1 = torch.zeros(big.param_floats, device=dev)
    big.backward(params, xd, None, ed, g1, ws, lambda_kl=1.0)
    g2 = torch.zeros_like(g1)
    big.forward(params, xd, None, ed, ws)
    big.loss(xd, 10.0, ws)
    big.backward(params, xd, None, ed, g2, ws, lambda_kl=1.0)
    assert torch.equal(g1, g2), "backward is not deterministic"
    assert torch.equal(dec, big.view(ws, "dec", (B, 80, T)))
    half = Plan(cfg, B // 2, T, lib=lib)
    wsh = torch.zeros(half.workspace_floats, device=dev)
    gsum = torch.zeros_like(g1)
    for i in range(2):
        sl = slice(i * B // 2, (i + 1) * B // 2)
        xs, es = xd[sl].contiguous(), ed[sl].contiguous()
        half.forward(params, xs, None, es, wsh)
        torch.testing.assert_close(half.view(wsh, "dec", (B // 2, 80, T)), dec[sl], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(half.view(wsh, "emb", (B // 2, 128)), emb[sl], rtol=1e-5, atol=1e-5)
        half.loss(xs, 10.0, wsh)
        gh = torch.zeros_like(g1)
        half.backward(params, xs, None, es, gh, wsh, lambda_kl=1.0)
        gsum += 0.5 * gh
    rel = ((gsum - g1).norm() / g1.norm()).item()
    # the two plans may tile/chunk a layer differently (different fp32 summation order), so a few of
    # the 1.5e8 ReLU decisions at this size flip between them (see branch_matched_oracle): 2e-3
    assert rel < 2e-3, rel


def rel_l2(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("kind,cfgname,B,T", [("emu", "tiny", 2, 32), pytest.param("gpu", "m80", 4, 128, marks=GPU)])
def test_bf16_compute_mode_vs_fp32_oracle(kind, cfgname, B, T):
    """compute_dtype "bf16r" (BASELINE config 3's precision on fp32 STORAGE; "bf16" itself is the bf16 storage engine,
    tests/test_graded_configs.py::test_bf16_storage_mode_at_graded_shape): conv / Linear operands rounded to bf16 inside the matrix core,
    fp32 accumulate, everything stored in fp32.  Tolerances: forward rel-L2 <= 3e-2 against the fp32
    oracle; whole gradient cosine >= 0.995 / rel-L2 <= 1e-1 against the oracle
    differentiated on the engine's own ReLU branch (no reference number exists for bf16 gradients;
    tests/test_model.py checks what matters, the loss curve)."""
    lib, dev = backend(kind)
    cfg = get_cfg(cfgname)
    sd = O.make_state_dict(cfg, 4)
    x, eps = O.make_inputs(cfg, B, T, 4)
    xd = x.to(dev)
    plan = Plan(cfg, B, T, lib=lib, compute_dtype="bf16r")
    assert plan.compute_dtype == "bf16r" and lib.avc_plan_compute_dtype(plan.h) == 1
    params = flat_params(plan, sd, dev)
    ws = torch.full((plan.workspace_floats,), float("nan"), device=dev)
    plan.forward(params, xd, None, eps.to(dev), ws)
    Tb, Cz = plan.latent_len, cfg["ContentEncoder"]["c_out"]
    muls = plan.view(ws, "muls", (B, 2 * Cz, Tb)).cpu()
    emb = plan.view(ws, "emb", (B, cfg["SpeakerEncoder"]["c_out"])).cpu()
    dec = plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len)).cpu()
    outs, grads_ref = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    e = {"emb": rel_l2(emb, outs["emb"]), "mu": rel_l2(muls[:, :Cz], outs["mu"]),
         "log_sigma": rel_l2(muls[:, Cz:], outs["log_sigma"]), "dec": rel_l2(dec, outs["dec"])}
    print(f"[{kind}/{cfgname}] bf16 forward rel-L2 vs fp32 oracle: {e}")
    assert max(e.values()) < 3e-2
    # the fp32 plan of the same shape must be untouched by the bf16 one
    plan32 = Plan(cfg, B, T, lib=lib)
    ws32 = torch.full((plan32.workspace_floats,), float("nan"), device=dev)
    plan32.forward(params, xd, None, eps.to(dev), ws32)
    torch.testing.assert_close(plan32.view(ws32, "dec", dec.shape).cpu(), outs["dec"], rtol=1e-4, atol=2e-5)
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    losses = plan.view(ws, "losses", (2,)).cpu()
    assert losses[0].item() == pytest.approx(outs["loss_rec"].item(), rel=2e-2)
    assert losses[1].item() == pytest.approx(outs["loss_kl"].item(), rel=5e-2)
    grads = torch.full((plan.param_floats,), float("nan"), device=dev)
    plan.backward(params, xd, None, eps.to(dev), grads, ws, lambda_kl=1.0)
    g = grads.cpu()
    assert torch.isfinite(g).all()

    def flat(gr):
        f = torch.zeros_like(g)
        for (off, n, shape), (k, v) in zip(plan.param_info, gr.items()):
            f[off:off + n] = v.reshape(-1)
        return f

    # (a) against the fp32 oracle differentiated on the ENGINE's ReLU branch: only the bf16 operand
    # rounding of the matrix products remains; (b) against the oracle's own branch: plus the units
    # that the perturbed forward pushed across a ReLU kink (reported, loosely bounded)
    _, grads_m = branch_matched_oracle(plan, ws, x, eps, sd, cfg)
    gm, gr = flat(grads_m), flat(grads_ref)
    cos_m = torch.nn.functional.cosine_similarity(g, gm, dim=0).item()
    cos_r = torch.nn.functional.cosine_similarity(g, gr, dim=0).item()
    print(f"[{kind}/{cfgname}] bf16 gradient vs fp32 oracle: same ReLU branch cosine {cos_m:.5f} rel-L2 {rel_l2(g, gm):.3e} | "
          f"oracle's own branch cosine {cos_r:.5f} rel-L2 {rel_l2(g, gr):.3e}")
    assert cos_m > 0.995 and rel_l2(g, gm) < 1e-1   # measured: 0.998 / 6e-2 on the tiny net, closer on the stock one
    assert cos_r > 0.98


@pytest.mark.parametrize("kind,cfgname,B,T", [("emu", "tiny", 3, 32), pytest.param("gpu", "m80", 33, 64, marks=GPU)])
def test_decoder_forward_as_two_half_batch_chains(kind, cfgname, B, T):
    """Large batches run the decoder forward as two half-batch kernel chains on two streams
    (engine.hip); every tensor is [B, ...] so the split is a pointer offset.  Same function."""
    lib, dev = backend(kind)
    cfg = get_cfg(cfgname)
    sd = O.make_state_dict(cfg, 2)
    x, eps = O.make_inputs(cfg, B, T, 2)
    plan = Plan(cfg, B, T, lib=lib)
    params = flat_params(plan, sd, dev)
    outs, _ = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    res = []
    for split_min in (10 ** 6, 2):
        plan = Plan(cfg, B, T, lib=lib, tuning={"dec_split_min": split_min})
        ws = torch.full((plan.workspace_floats,), float("nan"), device=dev)
        plan.forward(params, x.to(dev), None, eps.to(dev), ws)
        res.append(plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len)).cpu().clone())
        torch.testing.assert_close(res[-1], outs["dec"], rtol=1e-4, atol=2e-5)
        grads = torch.full((plan.param_floats,), float("nan"), device=dev)
        plan.loss(x.to(dev), 10.0, ws)
        plan.backward(params, x.to(dev), None, eps.to(dev), grads, ws, lambda_kl=1.0)   # consumes the split forward's saved tensors
        assert torch.isfinite(grads).all()
        res.append(grads.cpu().clone())
    torch.testing.assert_close(res[0], res[2], rtol=1e-5, atol=1e-6)
    assert ((res[1] - res[3]).norm() / res[1].norm()).item() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("part", ["decoder", "speaker"])
def test_partial_gradient_events_order_a_consumer_stream(part):
    """avc_plan_stream_wait_grads(AVC_GRADS_DECODER / AVC_GRADS_SPEAKER) lets a communication stream start on a bucket while
    the rest of the backward pass still runs.  A copy of that bucket issued on such a stream right behind the wait must already hold the
    FINAL gradients -- compared with the same range after a full device synchronisation (a too-early event would race with the
    weight-gradient / reduce launches that write the range)."""
    from adaptive_voice_conversion_amd import _lib
    lib, dev = backend("gpu")
    cfg = get_cfg("m80")
    sd = O.make_state_dict(cfg, 7)
    B, T = 32, 128
    x, eps = O.make_inputs(cfg, B, T, 7)
    plan = Plan(cfg, B, T, T, lib=None)
    params = flat_params(plan, sd, dev)
    ws = torch.zeros(plan.workspace_floats, device=dev)
    xd, ed = x.to(dev), eps.to(dev)
    kind = {"decoder": _lib.GRADS_DECODER, "speaker": _lib.GRADS_SPEAKER}[part]
    off, n = plan.param_range(kind)
    side = torch.cuda.Stream()
    for rep in range(3):
        grads = torch.full((plan.param_floats,), float("nan"), device=dev)
        early = torch.full((n,), float("nan"), device=dev)
        plan.forward(params, xd, None, ed, ws)
        plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
        plan.backward(params, xd, None, ed, grads, ws, lambda_kl=1.0)
        assert plan.stream_wait_grads(kind, side)
        with torch.cuda.stream(side):
            early.copy_(grads[off:off + n], non_blocking=True)
        torch.cuda.synchronize()
        assert torch.isfinite(early).all()
        assert torch.equal(early, grads[off:off + n])


@pytest.mark.parametrize("kind,cfgname,B,T", [("emu", "tiny", 2, 32), pytest.param("gpu", "m80", 32, 128, marks=pytest.mark.gpu)])
def test_two_plans_from_two_threads(kind, cfgname, B, T):
    """VERDICT r5 item 4c.  include/avc_hip.h: "two host threads need two plans"; since round 5 the helper streams are ONE set per device
    that every plan shares.  Two plans of different shapes, each driven by its own host thread on its own caller stream, several steps
    concurrently: every result must be bit-equal to the same plan run alone -- the shared streams may serialise the two, never mix them.
    Also: the side stream's priority is the first plan's (avc_plan_side_priority), whatever a later plan asks for."""
    import threading
    lib, dev = backend(kind)
    cfg = get_cfg(cfgname)
    sd = O.make_state_dict(cfg, 5)
    shapes = [(B, T), (B + 1, T + (8 if kind == "emu" else 64))]
    plans = [Plan(cfg, b, t, lib=lib, tuning=({"side_prio": 0} if i else None)) for i, (b, t) in enumerate(shapes)]
    assert lib.avc_plan_side_priority(plans[0].h) == lib.avc_plan_side_priority(plans[1].h)   # one side stream per device and process
    params = flat_params(plans[0], sd, dev)
    data = [tuple(t.to(dev) for t in O.make_inputs(cfg, b, t, 7 + i)) for i, (b, t) in enumerate(shapes)]
    streams = [torch.cuda.Stream(device=dev) if kind == "gpu" else None for _ in plans]

    def step(i):
        plan, (x, eps) = plans[i], data[i]
        ws = torch.zeros(plan.workspace_floats, device=dev)
        g = torch.zeros(plan.param_floats, device=dev)
        ctx = torch.cuda.stream(streams[i]) if streams[i] is not None else contextlib.nullcontext()
        with ctx:
            plan.forward(params, x, None, eps, ws)
            plan.loss(x, 10.0, ws)
            plan.backward(params, x, None, eps, g, ws, lambda_kl=1.0)
        if streams[i] is not None:
            streams[i].synchronize()
        return g.cpu(), plan.view(ws, "dec", (x.shape[0], cfg["Decoder"]["c_out"], plan.out_len)).cpu()

    import contextlib
    if kind == "gpu":
        torch.cuda.synchronize()
    alone = [step(0), step(1)]
    nsteps = 4 if kind == "gpu" else 1
    out = [[None] * nsteps, [None] * nsteps]
    err = []

    def worker(i):
        try:
            for k in range(nsteps):
                out[i][k] = step(i)
        except Exception as e:   # noqa: BLE001
            err.append(e)
    if kind == "gpu":
        th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
    else:   # the CPU simulator runs kernels on fibers of ONE host thread: its twin interleaves the two plans' steps instead
        for i in (1, 0):
            worker(i)
    assert not err, err
    for i in range(2):
        for k in range(nsteps):
            assert torch.equal(out[i][k][0], alone[i][0]), f"plan {i}, concurrent step {k}: gradients differ from the serial run"
            assert torch.equal(out[i][k][1], alone[i][1]), f"plan {i}, concurrent step {k}: dec differs from the serial run"


@pytest.mark.parametrize("kind,cfgname,B,T,compute,mode", [
    ("emu", "tiny", 2, 32, "fp32", "train"), ("emu", "tiny", 2, 32, "bf16s", "train"), ("emu", "tiny128", 2, 32, "fp32", "train"),
    ("emu", "tiny128", 2, 32, "bf16s", "train"), ("emu", "tiny8", 2, 64, "fp32", "inference"),
    pytest.param("gpu", "m80", 256, 128, "fp32", "train", marks=GPU), pytest.param("gpu", "m80", 256, 128, "bf16s", "train", marks=GPU),
    pytest.param("gpu", "m80", 4, 128, "fp32x3", "train", marks=GPU), pytest.param("gpu", "m80", 4, 128, "bf16r", "train", marks=GPU),
    pytest.param("gpu", "m512", 8, 128, "fp32", "train", marks=GPU), pytest.param("gpu", "m80", 64, 1024, "fp32", "train", marks=GPU),
    pytest.param("gpu", "m80", 1024, 128, "fp32", "inference", marks=GPU), pytest.param("gpu", "m80", 3, 40, "fp32", "train", marks=GPU)])
def test_one_launch_weight_pack_equals_the_per_image_kernels(kind, cfgname, B, T, compute, mode):
    """avc_plan_pack_weights is ONE launch over a device table; since round 6 its blocks read the state_dict tensors in contiguous runs into
    LDS and gather the image order from there (csrc/conv_gemm.hip: pack_staged).  Every byte it writes must equal what the op-level gather
    kernel writes image by image (`dbg_streams` bit 4 selects that path) -- over every image type of a plan: fp32 / bf16-pair 16-byte
    images, forward and transposed tap-flipped input-gradient images, the stacked AdaIN affines and their bias rows, plain images, the
    split-bf16 images of fp32x3 (legacy pieces), the 1104-channel in_conv whose input gradient covers only its first 1024 rows."""
    lib, dev = backend(kind)
    cfg = get_cfg(cfgname)
    sd = O.make_state_dict(cfg, 11)
    out = []
    for tuning in (None, {"dbg_streams": 16}):
        plan = Plan(cfg, B, T, lib=lib, compute_dtype=compute, mode=mode, tuning=tuning)
        params = flat_params(plan, sd, dev)
        ws = torch.full((plan.workspace_floats,), -3.0, device=dev)   # (bytes a path does not write stay -3: both must write the same set)
        plan.pack_weights(params, ws)
        if kind == "gpu":
            torch.cuda.synchronize()
        out.append(ws.view(torch.int32).cpu())
        plan.close()
    assert out[0].numel() == out[1].numel()
    written = int((out[1] != torch.tensor(-3.0).view(torch.int32)).sum())
    assert written > 0
    assert torch.equal(out[0], out[1]), f"{int((out[0] != out[1]).sum())} of {written} packed dwords differ"
