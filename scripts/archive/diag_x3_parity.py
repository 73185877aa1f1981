"""Diagnostic (GPU): per-tensor gradient error vs the fp64 oracle on the engine's ReLU branch at B=256, T=128, for the exact-fp32 engine
and the fp32x3 mode; prints the five worst tensors of each and whether the x3 plan really carries x3 images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_graded_configs as G
from oracle import avc_oracle as O
from adaptive_voice_conversion_amd.engine import Plan
from adaptive_voice_conversion_amd import _lib

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 128
for mode in ("fp32", "fp32x3"):
    G._COMPUTE[0] = mode
    cfg, sd, x, eps, plan, ws, out, grads = G._fwd_bwd("gpu", "m80", B, T)
    base = Plan(cfg, B, T, lib=_lib.load())
    print(mode, "workspace floats", plan.workspace_floats, "vs plain", base.workspace_floats, flush=True)
    masks = [m.cpu() for m in plan.relu_masks(ws)]
    sd64 = {k: v.double() for k, v in sd.items()}
    with O.relu_masks(masks):
        _, g64 = O.loss_and_grads(x.double(), eps.double(), sd64, cfg, 1.0)
    g = grads.cpu().double()
    errs = []
    for (off, n, shape), k in zip(plan.param_info, g64):
        d = g64[k].norm().item()
        if d > 1e-6:
            errs.append(((g[off:off + n].view(shape) - g64[k]).norm().item() / d, k))
    errs.sort(reverse=True)
    print(mode, "worst:", [(f"{e:.2e}", k) for e, k in errs[:5]], "median %.2e" % errs[len(errs) // 2][0], flush=True)
