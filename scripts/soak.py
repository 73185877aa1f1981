"""Soak: N train steps on fresh random batches, fp32 and bf16; losses must stay finite and go down (GPU only)."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from adaptive_voice_conversion_amd.config import default_config
from adaptive_voice_conversion_amd.solver import Solver
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for dtype in ("fp32", "bf16"):
    cfg = default_config(80)
    cfg["compute_dtype"] = dtype
    torch.manual_seed(0)
    s = Solver(cfg, SimpleNamespace())
    g = torch.Generator(device="cpu").manual_seed(1)
    base = torch.randn(64, 80, 128, generator=g).to(dev)          # a small "corpus": the model can fit it
    hist = []
    t0 = time.perf_counter()
    for it in range(N):
        idx = torch.randint(0, 64, (256,), generator=g).to(dev)
        x = base[idx] + 0.05 * torch.randn(256, 80, 128, device=dev)
        m = s.ae_step(x, s.kl_weight(it), sync=(it % 50 == 0 or it == N - 1))
        if it % 50 == 0 or it == N - 1:
            hist.append((it, round(m["loss_rec"], 4), round(m["loss_kl"], 3), round(m["grad_norm"], 3)))
    torch.cuda.synchronize()
    ok = all(v == v and abs(v) < 1e6 for h in hist for v in h[1:])
    print(dtype, "finite" if ok else "NOT FINITE", f"{(time.perf_counter()-t0)/N*1e3:.2f} ms/step incl. batch synthesis", hist, flush=True)
    assert ok and hist[-1][1] < hist[0][1]
