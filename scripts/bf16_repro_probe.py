"""Is the bf16 storage engine bit-reproducible ACROSS processes?  (bench lines of the same build showed final losses that differ in the
4th digit from run to run while the fp32 engine's are identical.)  One forward + loss + backward on fixed inputs; prints a checksum per
output and per gradient tensor.  Run it several times and diff the outputs.
usage: python scripts/bf16_repro_probe.py [dtype] [B] > out.txt"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd.engine import Plan
from adaptive_voice_conversion_amd.config import default_config
from oracle import avc_oracle as O   # (inputs / weights only: test infrastructure, this is a diagnostic script)

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16s"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = 128
dev = torch.device("cuda", 0)
cfg = O.stock_config(80)
sd = O.make_state_dict(cfg, 0)
x, eps = O.make_inputs(cfg, B, T, 0)
plan = Plan(cfg, B, T, compute_dtype=dtype)
flat = torch.zeros(plan.param_floats, device=dev)
for (off, n, shape), v in zip(plan.param_info, sd.values()):
    flat[off:off + n] = v.reshape(-1).to(dev)
h = lambda t: hashlib.md5(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]
for rep in range(2):
    ws = torch.full((plan.workspace_floats,), float("nan"), device=dev) if rep == 0 else torch.zeros(plan.workspace_floats + 1024, device=dev)[1024:]   # (another address and fill)
    xd, ed = x.to(dev), eps.to(dev)
    plan.forward(flat, xd, None, ed, ws)
    plan.loss(xd, cfg["lambda"]["lambda_rec"], ws)
    grads = torch.full((plan.param_floats,), float("nan"), device=dev)
    plan.backward(flat, xd, None, ed, grads, ws, lambda_kl=1.0)
    torch.cuda.synchronize()
    Cz = cfg["ContentEncoder"]["c_out"]
    print(f"rep{rep} muls", h(plan.view(ws, "muls", (B, 2 * Cz, plan.latent_len))))
    print(f"rep{rep} emb", h(plan.view(ws, "emb", (B, 128))))
    print(f"rep{rep} dec", h(plan.view(ws, "dec", (B, 80, plan.out_len))))
    print(f"rep{rep} losses", h(plan.view(ws, "losses", (2,))), plan.view(ws, "losses", (2,)).tolist())
    for (off, n, shape), k in zip(plan.param_info, sd):
        print(f"rep{rep} grad {k}", h(grads[off:off + n]))
