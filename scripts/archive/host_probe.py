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

# steady-state host cost per segment of ae_step (no synchronisation inside the loop: the GPU is busy, as in training)
for Bp in (256, 4):
    xb = torch.randn(Bp, 80, T, device=dev); eb = torch.randn(Bp, 128, T // 8, device=dev)
    planb, wsb = model._plan(Bp, T, T, dev)
    for _ in range(3): s.ae_step(xb, 1.0, eps=eb, sync=False)
    torch.cuda.synchronize()
    acc = {}
    def seg(name, fn):
        t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    n = 30
    t00 = time.perf_counter()
    for _ in range(n):
        fl = seg("flat_parameters", lambda: model.flat_parameters())
        pw = seg("_plan", lambda: model._plan(Bp, T, T, dev))
        gr = seg("flat_grads", lambda: model.flat_grads())
        seg("forward", lambda: planb.forward(fl, xb, None, eb, wsb))
        seg("loss", lambda: planb.loss(xb, 10.0, wsb))
        seg("backward", lambda: planb.backward(fl, xb, None, eb, gr, wsb, lambda_kl=1.0))
        seg("opt.step", lambda: s.opt.step(5.0))
    t01 = time.perf_counter()
    torch.cuda.synchronize()
    t02 = time.perf_counter()
    print(f"B={Bp}: host {1e3*(t01-t00)/n:.3f} ms/step, total {1e3*(t02-t00)/n:.3f} ms/step; " + ", ".join(f"{k} {1e3*v/n:.3f}" for k, v in acc.items()), flush=True)
