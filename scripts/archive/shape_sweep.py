"""Randomised shape sweep on the GPU: engine forward / loss / backward vs the oracle (branch-matched
gradients) for awkward (B, T) combinations.  Uses the helpers of tests/test_engine.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd.engine import Plan
from oracle import avc_oracle as O
from tests.test_engine import flat_params, check_grads, branch_matched_oracle
from tests.emu_util import backend
lib, dev = backend("gpu")
cfg = O.stock_config(80)
sd = O.make_state_dict(cfg, 3)
worst_all = 0.0
for B, T in [(1, 17), (3, 23), (5, 40), (2, 72), (33, 32), (4, 136), (2, 200), (7, 64), (1, 333), (36, 48)]:
    x, eps = O.make_inputs(cfg, B, T, B * 100 + T)
    plan = Plan(cfg, B, T, lib=lib)
    params = flat_params(plan, sd, dev)
    ws = torch.full((plan.workspace_floats,), float("nan"), device=dev)
    plan.forward(params, x.to(dev), None, eps.to(dev), ws)
    dec = plan.view(ws, "dec", (B, 80, plan.out_len)).cpu()
    outs, _ = O.loss_and_grads(x, eps, sd, cfg, 1.0) if plan.out_len == T else (O.ae_forward(x, eps, sd, cfg), None)
    ref_dec = outs["dec"] if isinstance(outs, dict) else outs[3]
    torch.testing.assert_close(dec, ref_dec, rtol=1e-4, atol=3e-5)
    msg = f"B={B} T={T} T'={plan.out_len}: forward ok"
    if plan.out_len == T:
        plan.loss(x.to(dev), 10.0, ws)
        grads = torch.full((plan.param_floats,), float("nan"), device=dev)
        plan.backward(params, x.to(dev), None, eps.to(dev), grads, ws, lambda_kl=1.0)
        _, grads_m = branch_matched_oracle(plan, ws, x, eps, sd, cfg)
        illc = T <= 24
        worst, med, total = check_grads(plan, grads, grads_m, tol=5e-3 if illc else 2e-4, cfg=cfg, zero_abs=2e-5)
        worst_all = max(worst_all, 0 if illc else worst)
        msg += f"; grads worst tensor {worst:.2e} median {med:.2e} whole {total:.2e}"
    print(msg, flush=True)
print("sweep ok; worst well-conditioned per-tensor rel-L2", f"{worst_all:.2e}")
