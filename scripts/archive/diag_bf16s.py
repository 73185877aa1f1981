"""GPU-box diagnostic: accuracy of the bf16 compute modes against the oracle twins (forward rel-L2, per-tensor gradient error distribution,
zero-gradient biases, whole-gradient cosine).  usage: python scripts/diag_bf16s.py gpu m80 256 128 bf16r bf16s"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import avc_oracle as O
import tests.test_graded_configs as G
from tests.test_engine import zero_grad_bias
kind, cfgname, B, T = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
for mode in sys.argv[5:]:
    G._COMPUTE[0] = mode
    cfg, sd, x, eps, plan, ws, out, grads = G._fwd_bwd(kind, cfgname, B, T)
    Cz = cfg["ContentEncoder"]["c_out"]
    mine = {"emb": out["emb"], "mu": out["muls"][:, :Cz], "log_sigma": out["muls"][:, Cz:], "dec": out["dec"]}
    o32 = dict(zip(("mu", "log_sigma", "emb", "dec"), O.ae_forward(x, eps, sd, cfg)))
    print(mode, "fwd vs fp32", {k: f"{G._rel(mine[k], o32[k]):.2e}" for k in mine})
    with O.bf16_operands():
        o16 = dict(zip(("mu", "log_sigma", "emb", "dec"), O.ae_forward(x, eps, sd, cfg)))
    print(mode, "fwd vs twin", {k: f"{G._rel(mine[k], o16[k]):.2e}" for k in mine})
    masks = [m.cpu() for m in plan.relu_masks(ws)]
    sd64 = {k: v.double() for k, v in sd.items()}
    with O.relu_masks(masks), O.bf16_operands():
        _, g64 = O.loss_and_grads(x.double(), eps.double(), sd64, cfg, 1.0)
    with O.relu_masks(masks):
        _, g64x = O.loss_and_grads(x.double(), eps.double(), sd64, cfg, 1.0)
    g = grads.cpu().double()
    errs, errsx, zb = [], [], []
    for (off, n, shape), k in zip(plan.param_info, g64):
        gi, ref = g[off:off + n].view(shape), g64[k]
        d = ref.norm().item()
        if zero_grad_bias(k, cfg):
            zb.append(((gi - ref).norm().item(), k)); continue
        errs.append(((gi - ref).norm().item() / d, k))
        errsx.append(((gi - g64x[k]).norm().item() / g64x[k].norm().item(), k))
    errs.sort(reverse=True); errsx.sort(reverse=True); zb.sort(reverse=True)
    print(mode, "grad vs twin64: worst", [(f"{e:.2e}", k) for e, k in errs[:4]], "median", f"{errs[len(errs)//2][0]:.2e}")
    print(mode, "grad vs exact64: worst", [(f"{e:.2e}", k) for e, k in errsx[:4]], "median", f"{errsx[len(errsx)//2][0]:.2e}")
    print(mode, "zero-grad biases abs err worst", [(f"{e:.2e}", k) for e, k in zb[:3]])
    gt = torch.cat([v.reshape(-1) for v in g64x.values()])
    print(mode, "whole-gradient cosine vs exact", torch.nn.functional.cosine_similarity(g[:0].new_tensor([0]) if False else torch.cat([g[off:off+n] for (off,n,shape) in plan.param_info]), gt, dim=0).item())
