"""GPU diagnostic: per-tensor gradient error of the engine vs the branch-matched oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd.engine import Plan
from adaptive_voice_conversion_amd import _lib
from oracle import avc_oracle as O

lib = _lib.load()
dev = torch.device("cuda", 0)

def run(cfgname, B, T, seed):
    cfg = {"tiny": O.tiny_config, "m80": lambda: O.stock_config(80)}[cfgname]()
    sd = O.make_state_dict(cfg, seed)
    x, eps = O.make_inputs(cfg, B, T, seed)
    xd = x.to(dev)
    plan = Plan(cfg, B, T, lib=lib)
    flat = torch.zeros(plan.param_floats)
    for (off, n, shape), (k, v) in zip(plan.param_info, sd.items()):
        flat[off:off + n] = v.reshape(-1)
    params = flat.to(dev)
    for rep in range(2):
        ws = torch.full((plan.workspace_floats,), float("nan") if rep == 0 else 0.0, device=dev)
        plan.forward(params, xd, None, eps.to(dev), ws)
        plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
        grads = torch.zeros(plan.param_floats, device=dev)
        plan.backward(params, xd, None, eps.to(dev), grads, ws, lambda_kl=1.0)
        masks = [m.cpu() for m in plan.relu_masks(ws)]
        with O.relu_masks(masks):
            o32, g32 = O.loss_and_grads(x, eps, sd, cfg, 1.0)
        gc = grads.cpu()
        print(f"=== {cfgname} B={B} T={T} seed={seed} rep={rep}")
        for (off, n, shape), k in zip(plan.param_info, sd.keys()):
            gi = gc[off:off + n].view(shape)
            d = g32[k].norm().item()
            e = (gi - g32[k]).norm().item()
            if d > 1e-6 and e / d > 2e-5:
                print(f"  {e/d:.2e} |g|={d:.2e} {k}")
        sys.stdout.flush()

run("m80", 3, 24, 4)
run("m80", 3, 48, 4)
