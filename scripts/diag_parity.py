"""GPU diagnostic: per-tensor gradient error of the engine vs the oracle in fp32 AND fp64."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd.engine import Plan
from adaptive_voice_conversion_amd import _lib
from oracle import avc_oracle as O

lib = _lib.load()
dev = torch.device("cuda", 0)

def run(cfgname, B, T, seed, transposed):
    cfg = {"tiny": O.tiny_config, "m80": lambda: O.stock_config(80), "m512": lambda: O.stock_config(512)}[cfgname]()
    sd = O.make_state_dict(cfg, seed)
    x, eps = O.make_inputs(cfg, B, T, seed)
    xd = x.to(dev)
    if transposed:
        xd = xd.transpose(1, 2).contiguous().transpose(1, 2)
    plan = Plan(cfg, B, T, lib=lib)
    flat = torch.zeros(plan.param_floats)
    for (off, n, shape), (k, v) in zip(plan.param_info, sd.items()):
        flat[off:off + n] = v.reshape(-1)
    params = flat.to(dev)
    ws = torch.zeros(plan.workspace_floats, device=dev)
    plan.forward(params, xd, None, eps.to(dev), ws)
    Tb, Cz = plan.latent_len, cfg["ContentEncoder"]["c_out"]
    muls = plan.view(ws, "muls", (B, 2 * Cz, Tb)).cpu()
    emb = plan.view(ws, "emb", (B, cfg["SpeakerEncoder"]["c_out"])).cpu()
    dec = plan.view(ws, "dec", (B, cfg["Decoder"]["c_out"], plan.out_len)).cpu()
    o32, g32 = O.loss_and_grads(x, eps, sd, cfg, 1.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    o64, g64 = O.loss_and_grads(x.double(), eps.double(), sd64, cfg, 1.0)
    print(f"=== {cfgname} B={B} T={T} seed={seed} transposed={transposed}")
    for name, mine, k in (("emb", emb, "emb"), ("mu", muls[:, :Cz], "mu"), ("ls", muls[:, Cz:], "log_sigma"), ("dec", dec, "dec")):
        print(f"  fwd {name:4s} max|gpu-o64| {(mine.double()-o64[k]).abs().max().item():.2e}   max|o32-o64| {(o32[k].double()-o64[k]).abs().max().item():.2e}")
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    grads = torch.zeros(plan.param_floats, device=dev)
    plan.backward(params, xd, None, eps.to(dev), grads, ws, lambda_kl=1.0)
    gc = grads.cpu()
    rows = []
    for (off, n, shape), k in zip(plan.param_info, sd.keys()):
        gi = gc[off:off + n].view(shape).double()
        d = g64[k].norm().item()
        e_gpu = (gi - g64[k]).norm().item()
        e_o32 = (g32[k].double() - g64[k]).norm().item()
        rows.append((e_gpu / max(d, 1e-30), e_o32 / max(d, 1e-30), d, k))
    rows = [r for r in rows if r[2] > 1e-6]
    rows.sort(reverse=True)
    for r in rows[:14]:
        print(f"  grad relL2 gpu-vs-o64 {r[0]:.2e}   o32-vs-o64 {r[1]:.2e}   |g|={r[2]:.2e}  {r[3]}")
    sys.stdout.flush()

for args in (("m80", 4, 128, 1, False), ("m80", 2, 256, 4, False), ("m80", 4, 128, 4, True), ("m80", 2, 128, 0, False)):
    run(*args)
