"""Which backward TEMPORARY of the bf16 storage engine is not reproducible?  Runs forward + loss + backward twice on one plan and compares the
dy arena (every dy tensor a weight-gradient launch reads has its own slot there, in issue order) and the named gradient buffers block by block.
usage: python scripts/bf16_repro_probe3.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd.engine import Plan
from oracle import avc_oracle as O   # (inputs / weights only: diagnostic script)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tun = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[2:]} or None
dev = torch.device("cuda", 0)
cfg = O.stock_config(80)
sd = O.make_state_dict(cfg, 0)
x, eps = O.make_inputs(cfg, B, 128, 0)
plan = Plan(cfg, B, 128, compute_dtype="bf16s", tuning=tun)
flat = torch.zeros(plan.param_floats, device=dev)
for (off, n, shape), v in zip(plan.param_info, sd.values()):
    flat[off:off + n] = v.reshape(-1).to(dev)
xd, ed = x.to(dev), eps.to(dev)
ws = torch.zeros(plan.workspace_floats, device=dev)
off = {k: plan.lib.avc_plan_buffer(plan.h, k.encode()) for k in ("wgrad_slab", "dy_arena", "g_tmp", "d_z", "d_muls", "d_emb", "d_cond", "d_dec")}
print("offsets:", off, "workspace floats:", plan.workspace_floats)
snaps = []
for rep in range(6):
    plan.forward(flat, xd, None, ed, ws)
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    grads = torch.zeros(plan.param_floats, device=dev)
    plan.backward(flat, xd, None, ed, grads, ws, lambda_kl=1.0)
    torch.cuda.synchronize()
    snaps.append(ws.view(torch.int32).clone())
# arena slots in issue order (engine.hip: avc_backward_impl), pair rows: C = 64
C = 64
Td = [16, 32, 32, 64, 64, 128, 128]
Te = [128, 128, 64, 64, 32, 32, 16]
slots = []
for l in range(5, -1, -1):
    slots += [(f"dec dy2[{l}]", B * C * Td[l + 1]), (f"dec dy1[{l}]", B * C * Td[l])]
slots += [("dec dy0", B * C * 16)]
slots += [(f"spk dense dz[{i}]", 128 * B) for i in range(12)]
slots += [("enc dyA(head)", B * C * 16)]
for l in range(5, -1, -1):
    slots += [(f"enc dyB[{l}] (dy of first conv)", B * C * Te[l]), (f"enc dyA[{l}] (dy of second conv of block {l - 1} / in_conv)", B * C * Te[l])]
slots += [("spk dyA(pool)", B * C * 16)]
for l in range(5, -1, -1):
    slots += [(f"spk dyB[{l}]", B * C * Te[l]), (f"spk dyA[{l}]", B * C * Te[l])]
r64 = lambda n: (n + 63) // 64 * 64
base = off["dy_arena"]
for a, b, lab in ((0, 1, "run 1 vs 2"), (1, 2, "run 2 vs 3"), (2, 3, "run 3 vs 4"), (3, 4, "run 4 vs 5"), (4, 5, "run 5 vs 6")):
    print(lab)
    shown = False
    o = base
    for name, n in slots:
        d = (snaps[a][o:o + n] != snaps[b][o:o + n])
        if d.any():
            idx = d.nonzero().flatten()
            rows = torch.unique(idx // 128 if "128" in name else idx)   # (coarse)
            print(f"   arena slot {name} (+{o - base}, {n} dwords): {int(d.sum())} dwords differ; first at {int(idx[0])}, last at {int(idx[-1])}")
            if not shown:
                shown = True
                va, vb = snaps[a][o:o + n], snaps[b][o:o + n]
                lo = lambda v: (v << 16).view(torch.float32)
                hi = lambda v: (v & -65536).view(torch.float32)
                T = int(name.split("T=")[1]) if "T=" in name else 128
                for i in idx[:40].tolist():
                    print(f"      dword {i}: sample {i // (C * T)} pair-row {(i // T) % C} frame {i % T}: lo {lo(va[i:i+1]).item():+.5e} vs {lo(vb[i:i+1]).item():+.5e}   hi {hi(va[i:i+1]).item():+.5e} vs {hi(vb[i:i+1]).item():+.5e}")
                rows = torch.unique(idx // T)
                print(f"      rows (sample*C + pair-row) touched: {rows.tolist()[:64]}  ({len(rows)} rows)")
        o += r64(n)
    for k in ("d_z", "d_muls", "d_emb", "d_cond", "d_dec", "g_tmp"):
        n = {"d_z": B * 128 * 16, "d_muls": B * 256 * 16, "d_emb": B * 128, "d_cond": B * 3072, "d_dec": B * 80 * 128, "g_tmp": 6 * B * C * 128}[k]
        d = snaps[a][off[k]:off[k] + n] != snaps[b][off[k]:off[k] + n]
        if d.any():
            print(f"   {k}: {int(d.sum())} of {n} dwords differ")
    d = snaps[a] != snaps[b]
    idx = d.nonzero().flatten()
    print(f"   whole workspace: {int(d.sum())} dwords differ; first at {int(idx[0]) if len(idx) else -1} (arena base {base}, slab base {off['wgrad_slab']})")
