"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference, read-only) in this container.  The reference cannot travel to
the GPU box, so its outputs are committed as small fixtures; inputs and weights
are regenerated from seeds by ``oracle.avc_oracle.make_state_dict/make_inputs``
(identical torch build on both machines) and guarded by checksums stored here.

Run:  python oracle/make_golden.py [out_dir]     (needs /root/reference; default out_dir = tests/golden)

tests/test_oracle_golden.py::test_committed_fixtures_equal_a_fresh_regeneration re-runs this recipe into
a scratch directory whenever /root/reference is present and compares it with the committed files.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import avc_oracle as O  # noqa: E402

REF = "/root/reference"
# Seeds found by `python oracle/make_golden.py --find-seed`: the first whose forward pass keeps EVERY ReLU pre-activation of
# the reference at least 4e-6 (stock 80-mel, B=1, T=64: 3e5 sites; found 5.1e-6) / 2e-5 (tiny) away from 0.  Two correct fp32
# implementations then take the same piecewise-linear branch, so the complete gradient tensors of these fixtures can be
# compared at 1e-4 with NO branch matching.  (At B=2, T=128 -- 1.2e6 sites -- the closest site is ~5e-7 for the best of
# 400 seeds: that fixture records the reference's decisions instead, see run_case.)  run_case asserts the margins.
M80_FULL_SEED, M80_FULL_MARGIN = 19, 4e-6
TINY_B2_SEED, TINY_MARGIN = 5, 2e-5
TINY_LRELU_SEED = 3
OUT_DIR = os.path.join(ROOT, "tests", "golden")


def import_reference():
    tb = types.ModuleType("tensorboardX")  # utils.py:3 (logging only, not installed)
    tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda s, *a, **k: None})
    sys.modules["tensorboardX"] = tb
    sys.modules["editdistance"] = types.ModuleType("editdistance")  # utils.py:4, unused
    sys.path.insert(0, REF)
    import model  # noqa
    return model


def sample_idx(n, k=16):
    g = np.random.RandomState(12345 + n % 9973)
    return np.sort(g.choice(n, size=min(k, n), replace=False))


def tensor_stats(t):
    a = t.detach().double().reshape(-1).numpy()
    idx = sample_idx(a.size)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum(), np.abs(a).max()], a[idx]])


FULL_GRADS_M80 = [  # complete gradient tensors kept in train_m80_t128_b2 (besides every bias): one block of each network
    "speaker_encoder.first_conv_layers.1.weight", "speaker_encoder.second_conv_layers.1.weight",      # stride-2 block, no norm
    "content_encoder.first_conv_layers.1.weight", "content_encoder.second_conv_layers.1.weight",      # stride-2 block, InstanceNorm
    "decoder.first_conv_layers.0.weight", "decoder.second_conv_layers.0.weight",                      # pixel-shuffle block, AdaIN
    "decoder.conv_affine_layers.0.weight", "speaker_encoder.first_dense_layers.0.weight",
]


class ReluRecorder:
    """Pre-activations of every ReLU the reference executes, in call order (forward pre-hooks on the networks'
    shared `act` modules, model.py:93-99; speaker encoder -> content encoder -> decoder = the order of
    oracle.avc_oracle.relu_masks and of avc_plan_relu_site)."""

    def __init__(self, ref):
        self.pre, self.on = [], False
        self.handles = [m.register_forward_pre_hook(self._hook) for m in ref.modules() if isinstance(m, (torch.nn.ReLU, torch.nn.LeakyReLU))]

    def _hook(self, mod, inp):
        if self.on:
            self.pre.append(inp[0].detach().clone())

    def close(self):
        for h in self.handles:
            h.remove()


def relu_margin(model_mod, cfg, B, T, seed):
    """smallest |pre-activation| over all ReLU sites of the reference's forward pass for that seed"""
    torch.manual_seed(0)
    ref = model_mod.AE(cfg)
    ref.load_state_dict(O.make_state_dict(cfg, seed), strict=True)
    x, eps = O.make_inputs(cfg, B, T, seed)
    rec = ReluRecorder(ref)
    rec.on = True
    with torch.no_grad():
        emb = ref.speaker_encoder(x)
        mu, ls = ref.content_encoder(x)
        ref.decoder(mu + torch.exp(ls / 2) * eps, emb)
    rec.close()
    return min(float(p.abs().min()) for p in rec.pre)


def find_margin_seed(model_mod, cfg, B, T, margin, first=0, tries=400):
    """first seed >= `first` whose forward pass keeps every ReLU pre-activation at least `margin` away from 0:
    two correct fp32 implementations then take the SAME piecewise-linear branch, and complete gradient tensors can
    be compared at 1e-4 without any branch matching."""
    for seed in range(first, first + tries):
        m = relu_margin(model_mod, cfg, B, T, seed)
        if m >= margin:
            return seed, m
    raise RuntimeError("no seed with that margin")


def run_case(model_mod, name, cfg, B, T, seed, n_steps, full_outputs=True, full_grads=None, relu_record=False, margin=0.0):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = model_mod.AE(cfg)
    spec = O.param_spec(cfg)
    ref_sd = ref.state_dict()
    assert [k for k, _ in spec] == list(ref_sd.keys()), "param_spec order != reference state_dict"
    assert all(tuple(ref_sd[k].shape) == s for k, s in spec), "param_spec shapes != reference"
    sd = O.make_state_dict(cfg, seed)
    ref.load_state_dict(sd, strict=True)
    x, eps = O.make_inputs(cfg, B, T, seed)
    o = cfg["optimizer"]
    opt = torch.optim.Adam(ref.parameters(), lr=o["lr"], betas=(o["beta1"], o["beta2"]),
                           amsgrad=o["amsgrad"], weight_decay=o["weight_decay"])
    out = {"B": B, "T": T, "seed": seed, "n_steps": n_steps,
           "x_stats": tensor_stats(x), "eps_stats": tensor_stats(eps),
           "w_stats": np.stack([tensor_stats(v) for v in sd.values()])}
    names = [k for k, _ in spec]
    rec = ReluRecorder(ref) if relu_record else None
    for step in range(n_steps):
        lam = 1.0
        # --- solver.py:81-97 restated around the real reference AE, eps injected
        # exactly as model.py:380-385 composes the sub-modules
        if rec is not None:
            rec.on = step == 0
        emb = ref.speaker_encoder(x)
        mu, ls = ref.content_encoder(x)
        dec = ref.decoder(mu + torch.exp(ls / 2) * eps, emb)
        if rec is not None:
            rec.on = False
        loss_rec = torch.nn.L1Loss()(dec, x)
        loss_kl = 0.5 * torch.mean(torch.exp(ls) + mu ** 2 - 1 - ls)
        loss = cfg["lambda"]["lambda_rec"] * loss_rec + lam * loss_kl
        opt.zero_grad()
        loss.backward()
        if step == 0:
            if full_outputs:
                out["mu"] = mu.detach().numpy()
                out["log_sigma"] = ls.detach().numpy()
                out["emb"] = emb.detach().numpy()
                out["dec"] = dec.detach().numpy()
            out["out_stats"] = np.stack([tensor_stats(t) for t in (mu, ls, emb, dec)])
            grads = dict(ref.named_parameters())
            out["grad_stats"] = np.stack([tensor_stats(grads[k].grad) for k in names])
            # a few complete small gradients (biases + one dense weight)
            for k in ("speaker_encoder.output_layer.bias", "content_encoder.mean_layer.bias",
                      "decoder.conv_affine_layers.0.bias", "decoder.out_conv_layer.bias",
                      "speaker_encoder.conv_bank.0.bias", "decoder.first_conv_layers.0.bias"):
                out["grad/" + k] = grads[k].grad.detach().clone().numpy()
            if full_grads is not None:   # complete tensors: every bias + the listed weights ("all": everything)
                for k in names:
                    if full_grads == "all" or k.endswith(".bias") or k in full_grads:
                        out["gradfull/" + k] = grads[k].grad.detach().clone().numpy()
            if rec is not None:   # the reference's ReLU decisions (bit-packed, call order) and how far from a kink they were
                out["relu_bits"] = np.packbits(np.concatenate([(p > 0).reshape(-1).numpy() for p in rec.pre]))
                out["relu_sizes"] = np.array([p.numel() for p in rec.pre], dtype=np.int64)
                out["relu_margin"] = min(float(p.abs().min()) for p in rec.pre)
                assert out["relu_margin"] >= margin, (name, out["relu_margin"], margin)
                flat = np.concatenate([p.reshape(-1).numpy() for p in rec.pre])
                near = np.nonzero(np.abs(flat) < 2e-5)[0]     # sites a correct fp32 implementation may decide differently
                out["relu_near_idx"] = near.astype(np.int64)
                out["relu_near_pre"] = flat[near]
        gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=o["grad_norm"])
        opt.step()
        out[f"loss_rec_{step}"] = float(loss_rec)
        out[f"loss_kl_{step}"] = float(loss_kl)
        out[f"grad_norm_{step}"] = float(gn)
        psd = ref.state_dict()
        out[f"param_stats_{step}"] = np.stack([tensor_stats(psd[k]) for k in names])
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss_rec", out["loss_rec_0"], "loss_kl", out["loss_kl_0"], "gn", out["grad_norm_0"],
          os.path.getsize(path) // 1024, "KiB")


def run_inference_case(model_mod, name, cfg, Ts, Tc, seed):
    torch.manual_seed(0)
    ref = model_mod.AE(cfg)
    sd = O.make_state_dict(cfg, seed)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    x, _ = O.make_inputs(cfg, 1, Ts, seed)
    xc, _ = O.make_inputs(cfg, 1, Tc, seed + 7)
    with torch.no_grad():
        dec = ref.inference(x, xc)
        emb = ref.get_speaker_embeddings(xc)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, Ts=Ts, Tc=Tc, seed=seed, dec=dec.numpy(), emb=emb.numpy(),
                        x_stats=tensor_stats(x), xc_stats=tensor_stats(xc))
    print(name, tuple(dec.shape), os.path.getsize(path) // 1024, "KiB")


def make_init_golden(model_mod):
    """Default-initialisation pin: AE(config) of the reference under torch.manual_seed(0)
    (solver.py:72 builds the model with PyTorch's default Conv/Linear init)."""
    out = {}
    for name, cfg in (("m80", O.stock_config(80)), ("tiny", O.tiny_config())):
        torch.manual_seed(0)
        ref = model_mod.AE(cfg)
        out[name] = np.stack([tensor_stats(v) for v in ref.state_dict().values()])
    np.savez_compressed(os.path.join(OUT_DIR, "init_seed0.npz"), **out)


def main(out_dir=None):
    global OUT_DIR
    if out_dir:
        OUT_DIR = out_dir
    model_mod = import_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    c80 = O.stock_config(80)
    run_case(model_mod, "train_m80_t128_b2", c80, B=2, T=128, seed=0, n_steps=3, full_grads=FULL_GRADS_M80, relu_record=True)
    run_case(model_mod, "train_m80_t64_b1_full", c80, B=1, T=64, seed=M80_FULL_SEED, n_steps=1, full_grads=FULL_GRADS_M80, relu_record=True,
             margin=M80_FULL_MARGIN)
    run_case(model_mod, "train_m80_t128_b4_s1", c80, B=4, T=128, seed=1, n_steps=1, full_outputs=False)
    run_case(model_mod, "train_m80_t256_b1", c80, B=1, T=256, seed=2, n_steps=1, full_outputs=False)
    run_case(model_mod, "train_m512_t128_b1", O.stock_config(512), B=1, T=128, seed=3, n_steps=1, full_outputs=False)
    run_case(model_mod, "train_tiny_t32_b2", O.tiny_config(), B=2, T=32, seed=TINY_B2_SEED, n_steps=3, full_grads="all", relu_record=True, margin=TINY_MARGIN)
    run_case(model_mod, "train_tiny_t24_b3", O.tiny_config(), B=3, T=24, seed=5, n_steps=1)
    # act: lrelu (config-legal, model.py:93-99): every tensor of the tiny net, a margin seed again
    run_case(model_mod, "train_tiny_lrelu_t32_b2", O.tiny_config(act="lrelu"), B=2, T=32, seed=TINY_LRELU_SEED, n_steps=1, full_grads="all",
             relu_record=True, margin=TINY_MARGIN)
    run_inference_case(model_mod, "infer_m80_t100_c77", c80, Ts=100, Tc=77, seed=6)
    run_inference_case(model_mod, "infer_tiny_t37_c19", O.tiny_config(), Ts=37, Tc=19, seed=7)
    make_init_golden(model_mod)


if __name__ == "__main__" and "--find-seed" in sys.argv:
    mm = import_reference()
    print("m80 B=1 T=64:", find_margin_seed(mm, O.stock_config(80), 1, 64, M80_FULL_MARGIN, tries=3000))
    print("tiny B=2 T=32:", find_margin_seed(mm, O.tiny_config(), 2, 32, TINY_MARGIN, first=4))
    print("tiny lrelu B=2 T=32:", find_margin_seed(mm, O.tiny_config(act="lrelu"), 2, 32, TINY_MARGIN))
    sys.exit(0)

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
