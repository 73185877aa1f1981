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


def run_case(model_mod, name, cfg, B, T, seed, n_steps, full_outputs=True):
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
    for step in range(n_steps):
        lam = 1.0
        # --- solver.py:81-97 restated around the real reference AE, eps injected
        # exactly as model.py:380-385 composes the sub-modules
        emb = ref.speaker_encoder(x)
        mu, ls = ref.content_encoder(x)
        dec = ref.decoder(mu + torch.exp(ls / 2) * eps, emb)
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
    run_case(model_mod, "train_m80_t128_b2", c80, B=2, T=128, seed=0, n_steps=3)
    run_case(model_mod, "train_m80_t128_b4_s1", c80, B=4, T=128, seed=1, n_steps=1, full_outputs=False)
    run_case(model_mod, "train_m80_t256_b1", c80, B=1, T=256, seed=2, n_steps=1, full_outputs=False)
    run_case(model_mod, "train_m512_t128_b1", O.stock_config(512), B=1, T=128, seed=3, n_steps=1, full_outputs=False)
    run_case(model_mod, "train_tiny_t32_b2", O.tiny_config(), B=2, T=32, seed=4, n_steps=3)
    run_case(model_mod, "train_tiny_t24_b3", O.tiny_config(), B=3, T=24, seed=5, n_steps=1)
    run_inference_case(model_mod, "infer_m80_t100_c77", c80, Ts=100, Tc=77, seed=6)
    run_inference_case(model_mod, "infer_tiny_t37_c19", O.tiny_config(), Ts=37, Tc=19, seed=7)
    make_init_golden(model_mod)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
