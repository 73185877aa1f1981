"""Where does the bf16 storage engine's backward stop being bit-reproducible?  In ONE process: the same plan + workspace twice (A), then a
second workspace (B); per mode (default / single_stream / no decoder split / B) the number of gradient tensors whose bits differ and the
largest relative difference.  usage: python scripts/bf16_repro_probe2.py [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd.engine import Plan
from oracle import avc_oracle as O   # (inputs / weights only: diagnostic script)

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16s"
dev = torch.device("cuda", 0)
cfg = O.stock_config(80)
sd = O.make_state_dict(cfg, 0)
names = list(sd)


def run(plan, flat, xd, ed, ws):
    plan.forward(flat, xd, None, ed, ws)
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    grads = torch.full((plan.param_floats,), float("nan"), device=dev)
    plan.backward(flat, xd, None, ed, grads, ws, lambda_kl=1.0)
    torch.cuda.synchronize()
    return grads


def cmp(plan, g1, g2, label):
    bad, worst, wname = 0, 0.0, ""
    for (off, n, shape), k in zip(plan.param_info, names):
        a, b = g1[off:off + n], g2[off:off + n]
        if not torch.equal(a, b):
            bad += 1
            r = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
            if r > worst:
                worst, wname = r, k
    print(f"  {label}: {bad} of {len(names)} tensors differ; worst rel-L2 {worst:.2e} ({wname})", flush=True)
    if VERBOSE and bad:
        for (off, n, shape), k in zip(plan.param_info, names):
            a, b = g1[off:off + n], g2[off:off + n]
            if not torch.equal(a, b):
                print(f"      {k}: max|a-b| {(a - b).abs().max().item():.3e}  max|b| {b.abs().max().item():.3e}  elements differing {(a != b).sum().item()} of {n}")


VERBOSE = len(sys.argv) > 2
CASES = ((256, None), (256, {"single_stream": 1}), (256, {"dec_split_min": 100000}), (64, None), (16, None)) if not VERBOSE else ((16, None), (16, {"wgrad_batch": 1}), (16, {"side_prio": 0}), (256, {"dec_split_min": 100000}), (256, None))
if len(sys.argv) > 2 and sys.argv[2] == "knobs":
    VERBOSE = False
    CASES = tuple((16, t) for t in ({"dbg_streams": 5}, {"dbg_streams": 9}, {"dbg_streams": 13}, {"dbg_streams": 1})) if len(sys.argv) > 3 else tuple((16, t) for t in (None, {"dgrad_par": 0}, {"conv_in_fuse": 0}, {"wgrad_batch": 1000}, {"wgrad_batch": 4}, {"wgrad_batch_wgs": 32}, {"bank_switch": 0}, {"side_prio": 0},
                                     {"wgrad_ablation": 8}, {"bh_ck5": 16}))
for B, tuning in CASES:
    x, eps = O.make_inputs(cfg, B, 128, 0)
    plan = Plan(cfg, B, 128, compute_dtype=dtype, tuning=tuning)
    flat = torch.zeros(plan.param_floats, device=dev)
    for (off, n, shape), v in zip(plan.param_info, sd.values()):
        flat[off:off + n] = v.reshape(-1).to(dev)
    xd, ed = x.to(dev), eps.to(dev)
    print(f"{dtype} B={B} tuning={tuning}")
    ws = torch.zeros(plan.workspace_floats, device=dev)
    gs = [run(plan, flat, xd, ed, ws) for _ in range(8)]
    g1, g2, g3 = gs[0], gs[1], gs[2]
    cmp(plan, g1, g2, "same workspace, run 1 vs 2")
    cmp(plan, g2, g3, "same workspace, run 2 vs 3")
    nbad = sum(0 if torch.equal(gs[i], gs[i + 1]) else 1 for i in range(7))
    print(f"  consecutive runs that differ: {nbad} of 7")
    ws2 = torch.full((plan.workspace_floats,), float("nan"), device=dev)
    g4 = run(plan, flat, xd, ed, ws2)
    cmp(plan, g1, g4, "second workspace (NaN-filled) vs run 1")
