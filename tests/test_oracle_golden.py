"""Pins the CPU oracle (oracle/avc_oracle.py) to outputs of the REAL reference.

tests/golden/*.npz were produced by oracle/make_golden.py, which imports
/root/reference/model.py and drives it with a restated solver.py:81-97 step
(real torch.optim.Adam + clip_grad_norm_).  The reference has no tests of its
own, so these fixtures are the parity pins.
"""
import os

import numpy as np
import pytest
import torch

from oracle import avc_oracle as O


def _stats(t):
    from oracle.make_golden import tensor_stats
    return tensor_stats(t)


CASES = [
    ("train_m80_t128_b2", lambda: O.stock_config(80)),
    ("train_m80_t64_b1_full", lambda: O.stock_config(80)),
    ("train_m80_t128_b4_s1", lambda: O.stock_config(80)),
    ("train_m80_t256_b1", lambda: O.stock_config(80)),
    ("train_m512_t128_b1", lambda: O.stock_config(512)),
    ("train_tiny_t32_b2", O.tiny_config),
    ("train_tiny_t24_b3", O.tiny_config),
    ("train_tiny_lrelu_t32_b2", lambda: O.tiny_config(act="lrelu")),
]


def _close_stats(a, b, rtol, atol):
    # columns: l2, sum, absmax, then sampled entries
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name,cfgf", CASES)
def test_train_case_matches_reference(name, cfgf, golden_dir):
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = cfgf()
    B, T, seed, n_steps = int(g["B"]), int(g["T"]), int(g["seed"]), int(g["n_steps"])
    sd = O.make_state_dict(cfg, seed)
    x, eps = O.make_inputs(cfg, B, T, seed)
    # the regenerated inputs are the ones the reference saw
    np.testing.assert_allclose(_stats(x), g["x_stats"], rtol=0, atol=0)
    np.testing.assert_allclose(_stats(eps), g["eps_stats"], rtol=0, atol=0)
    np.testing.assert_allclose(np.stack([_stats(v) for v in sd.values()]), g["w_stats"], rtol=0, atol=0)
    names = [k for k, _ in O.param_spec(cfg)]
    assert len(names) == 2 * (len(names) // 2)
    opt = O.make_opt(sd, cfg)
    for step in range(n_steps):
        meta, outs, grads_clipped = O.ae_step(x, eps, sd, opt, cfg, 1.0)
        # Step 0 is a strict pin.  From step 1 on, Adam's first update is
        # sign-like (m/sqrt(v) = +-1), so elements whose gradient is ~0 +- fp
        # noise move by +-lr differently in any two implementations and the
        # trajectories decorrelate at the 1e-3 level (measured: reference vs
        # this oracle, 24 elements after step 0) -> loose tolerances there.
        tl, tg, tp = (2e-5, 1e-4, 2e-4) if step == 0 else (2e-3, 5e-3, 2e-3)
        assert meta["loss_rec"] == pytest.approx(float(g[f"loss_rec_{step}"]), rel=tl)
        assert meta["loss_kl"] == pytest.approx(float(g[f"loss_kl_{step}"]), rel=tl)
        assert meta["grad_norm"] == pytest.approx(float(g[f"grad_norm_{step}"]), rel=tg)
        if step == 0:
            if "dec" in g:
                for k in ("mu", "log_sigma", "emb", "dec"):
                    np.testing.assert_allclose(outs[k].numpy(), g[k], rtol=1e-4, atol=2e-5)
            ostats = np.stack([_stats(outs[k]) for k in ("mu", "log_sigma", "emb", "dec")])
            _close_stats(ostats, g["out_stats"], rtol=1e-4, atol=2e-5)
        pstats = np.stack([_stats(sd[k]) for k in names])
        ref = g[f"param_stats_{step}"]
        # per-tensor L2 norm of every parameter after the step
        np.testing.assert_allclose(pstats[:, 0], ref[:, 0], rtol=tp, atol=1e-6)
        # sampled entries: all but a handful (sign-flipped ~0-gradient elements, +-2*lr) agree
        bad = np.abs(pstats[:, 3:] - ref[:, 3:]) > (2e-6 + 1e-4 * np.abs(ref[:, 3:]))
        assert bad.mean() < (5e-3 if step == 0 else 0.5)
        assert np.abs(pstats[:, 3:] - ref[:, 3:]).max() < 4 * cfg["optimizer"]["lr"] * (step + 1)


def test_unclipped_gradients_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_tiny_t32_b2.npz"))
    cfg = O.tiny_config()
    sd = O.make_state_dict(cfg, int(g["seed"]))
    x, eps = O.make_inputs(cfg, int(g["B"]), int(g["T"]), int(g["seed"]))
    outs, grads = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    names = [k for k, _ in O.param_spec(cfg)]
    gs = np.stack([_stats(grads[k]) for k in names])
    ref = g["grad_stats"]
    # per-tensor L2 norms within 1e-4 relative (abs 1e-6 for the analytically-zero bias grads)
    np.testing.assert_allclose(gs[:, 0], ref[:, 0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gs[:, 3:], ref[:, 3:], rtol=2e-3, atol=2e-6)
    for k in g.files:
        if k.startswith("grad/"):
            np.testing.assert_allclose(grads[k[5:]].numpy(), g[k], rtol=1e-3, atol=2e-6)


@pytest.mark.parametrize("name,cfgf", [("train_m80_t64_b1_full", lambda: O.stock_config(80)), ("train_tiny_t32_b2", O.tiny_config),
                                       ("train_m80_t128_b2", lambda: O.stock_config(80)), ("train_tiny_lrelu_t32_b2", lambda: O.tiny_config(act="lrelu"))])
def test_complete_gradient_tensors_and_relu_decisions_match_reference(name, cfgf, golden_dir):
    """The fixtures with a ReLU record hold COMPLETE gradient tensors of the reference (every bias + one block of each
    network; everything for the tiny net) and its 0/1 ReLU decisions.  The two margin fixtures (no pre-activation
    within 4e-6 / 2e-5 of a kink) pin the restatement with NO branch matching: identical decisions, per-tensor
    rel-L2 <= 1e-4.  The B=2, T=128 fixture has sites ~1e-7 from a kink: there the restatement may only differ
    from the reference at recorded near-kink sites, and is compared on the reference's recorded branch."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = cfgf()
    seed = int(g["seed"])
    sd = O.make_state_dict(cfg, seed)
    x, eps = O.make_inputs(cfg, int(g["B"]), int(g["T"]), seed)
    log = []
    with O.relu_masks(None, log=log):
        O.ae_forward(x, eps, sd, cfg)
    assert [p.numel() for p in log] == list(g["relu_sizes"])
    n = int(g["relu_sizes"].sum())
    ref_bits = np.unpackbits(g["relu_bits"])[:n].astype(bool)
    mine = np.concatenate([(p > 0).reshape(-1).numpy() for p in log])
    diff = np.nonzero(mine != ref_bits)[0]
    margin_fixture = float(g["relu_margin"]) >= 4e-6
    if margin_fixture:
        assert diff.size == 0, diff[:10]
        _, grads = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    else:
        assert np.isin(diff, g["relu_near_idx"]).all(), "a ReLU decision differs from the reference's away from any kink"
        masks, o = [], 0
        for p in log:
            masks.append(torch.from_numpy(ref_bits[o:o + p.numel()].reshape(tuple(p.shape))))
            o += p.numel()
        with O.relu_masks(masks):
            _, grads = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    checked = 0
    for k in g.files:
        if not k.startswith("gradfull/"):
            continue
        ref, mine_g = torch.from_numpy(g[k]), grads[k[9:]]
        d = ref.norm().item()
        if d < 1e-5:      # analytically-zero bias gradients
            assert (mine_g - ref).abs().max().item() < 2e-6, k
        else:
            assert (mine_g - ref).norm().item() / d < 1e-4, (k, (mine_g - ref).norm().item() / d)
        checked += 1
    assert checked >= 60


@pytest.mark.parametrize("name,cfgf", [("infer_m80_t100_c77", lambda: O.stock_config(80)),
                                       ("infer_tiny_t37_c19", O.tiny_config)])
def test_inference_matches_reference(name, cfgf, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = cfgf()
    seed = int(g["seed"])
    sd = O.make_state_dict(cfg, seed)
    x, _ = O.make_inputs(cfg, 1, int(g["Ts"]), seed)
    xc, _ = O.make_inputs(cfg, 1, int(g["Tc"]), seed + 7)
    dec = O.ae_inference(x, xc, sd, cfg)
    emb = O.speaker_encoder(xc, sd, cfg)
    assert tuple(dec.shape) == tuple(g["dec"].shape)  # T' = 8*ceil(T/8) style growth, SURVEY §3.3
    np.testing.assert_allclose(dec.numpy(), g["dec"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=1e-4, atol=2e-5)


def test_short_input_raises_like_reference():
    cfg = O.stock_config(80)
    sd = O.make_state_dict(cfg, 0)
    x, eps = O.make_inputs(cfg, 1, 16, 0)  # T <= 16 -> the decoder's k=5 conv sees T_l = 2
    with pytest.raises(RuntimeError, match="Padding size"):
        O.ae_forward(x, eps, sd, cfg)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference only exists in the build container")
def test_committed_fixtures_equal_a_fresh_regeneration(golden_dir, tmp_path):
    """oracle/make_golden.py (which imports the REAL reference from /root/reference) re-run into a scratch
    directory reproduces every committed fixture bit for bit -- the goldens are the reference's outputs,
    and the committed recipe is the one that made them."""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "oracle", "make_golden.py"), str(tmp_path)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(golden_dir, "*.npz")))
    assert names and names == sorted(os.path.basename(f) for f in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    for n in names:
        a, b = np.load(os.path.join(golden_dir, n)), np.load(os.path.join(str(tmp_path), n))
        assert set(a.files) == set(b.files), n
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (n, k)


def test_aten_op_mode_of_the_oracle_is_the_same_function(golden_dir):
    """bench.py's cpu_baseline leg times the oracle with the reference's own ATen op choices
    (O.use_aten_ops): same values as the explicit restatement and as the reference's golden."""
    g = np.load(os.path.join(golden_dir, "train_m80_t128_b2.npz"))
    cfg = O.stock_config(80)
    sd = O.make_state_dict(cfg, int(g["seed"]))
    x, eps = O.make_inputs(cfg, int(g["B"]), int(g["T"]), int(g["seed"]))
    o1, g1 = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    O.use_aten_ops(True)
    try:
        o2, g2 = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    finally:
        O.use_aten_ops(False)
    np.testing.assert_allclose(o2["dec"].numpy(), g["dec"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(o2["dec"], o1["dec"], rtol=1e-5, atol=1e-5)
    assert float(o2["loss_rec"]) == pytest.approx(float(g["loss_rec_0"]), rel=1e-5)
    for k in g1:
        d = g1[k].norm().item()
        if d > 1e-6:
            assert (g1[k] - g2[k]).norm().item() / d < 1e-4, k
