"""How far does the host run ahead?  Enqueue time per train step vs. GPU time per step (GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from adaptive_voice_conversion_amd.config import default_config
from adaptive_voice_conversion_amd.solver import Solver

dev = torch.device("cuda", 0)
cfg = default_config(80)
torch.manual_seed(0)
s = Solver(cfg, SimpleNamespace())
B, T = 256, 128
x = torch.randn(B, 80, T, device=dev)
eps = torch.randn(B, 128, T // 8, device=dev)
for _ in range(3): s.ae_step(x, 1.0, eps=eps, sync=False)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): s.ae_step(x, 1.0, eps=eps, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/n:.3f} ms/step   total {1e3*(t2-t0)/n:.3f} ms/step", flush=True)
# phases
import ctypes
model = s.model; flat = model.flat_parameters(); plan, ws = model._plan(B, T, T, dev); g = model.flat_grads()
torch.cuda.synchronize()
for name, fn in (("forward", lambda: plan.forward(flat, x, None, eps, ws)), ("loss", lambda: plan.loss(x, 10.0, ws)),
                 ("backward", lambda: plan.backward(flat, x, None, eps, g, ws, lambda_kl=1.0))):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host {1e3*(t1-t0):.3f} ms, until done {1e3*(t2-t0):.3f} ms", flush=True)
