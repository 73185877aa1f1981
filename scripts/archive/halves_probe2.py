"""Follow-up diagnostics: (a) is a lone B = 128 plan (decoder as two 64-sample chains) deterministic?  (b) does a second ALIVE plan (more
HIP streams / events in the process) slow the step down even when it is never used?  (c) alternating between two whole-batch plans."""
import copy, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd.solver import Solver
from bench import stock_config
dev = torch.device("cuda", 0)
cfg = stock_config(80); cfg["pipeline_halves"] = False

def mk(B, T, tun=None):
    torch.manual_seed(0)
    s = Solver(copy.deepcopy(cfg), types.SimpleNamespace(store_model_path=None, load_model=False, data_dir=None, logdir="/tmp/avc_log", tuning=tun or {}))
    x = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(1)).to(dev)
    p = s.model._plan(B, T, T, dev)[0]
    eps = torch.randn(B, 128, p.latent_len, generator=torch.Generator().manual_seed(2)).to(dev)
    return s, x, eps

def timed(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n

for tun in ({}, {"dec_split_min": 256}, {"dec_split_min": 64}):
    outs = []
    for r in range(3):
        s, x, eps = mk(128, 128, tun)
        for _ in range(10): s.ae_step(x, 1.0, eps=eps, sync=False)
        torch.cuda.synchronize(); outs.append(s.model.flat_parameters().clone())
    print(f"(a) B=128 T=128 whole batch, tuning {tun}: runs differ in", [int((o != outs[0]).sum()) for o in outs[1:]], "elements")

s, x, eps = mk(256, 128)
t0 = timed(lambda: s.ae_step(x, 1.0, eps=eps, sync=False))
extra = [s.model._plan(128, 128, 128, dev, slot=k) for k in (0, 1)]
t1 = timed(lambda: s.ae_step(x, 1.0, eps=eps, sync=False))
print(f"(b) B=256 whole-batch step: {t0:.3f} ms alone, {t1:.3f} ms with two more (unused) plans alive")
pa = s.model._plan(256, 128, 128, dev); pb = s.model._plan(256, 128, 128, dev, slot=1)
flat, g = s.model.flat_parameters(), s.model.flat_grads()
def step(p, w):
    p.forward(flat, x, None, eps, w); p.loss(x, 10.0, w); p.backward(flat, x, None, eps, g, w, lambda_kl=1.0)
ta = timed(lambda: step(*pa)); tb = timed(lambda: (step(*pa), step(*pb)))
print(f"(c) forward+loss+backward on ONE whole-batch plan: {ta:.3f} ms; plan A then plan B (two passes): {tb:.3f} ms")
ha, hb = extra
xa, ea = x[:128], eps[:128]
th = timed(lambda: step(ha[0], ha[1]) if False else (ha[0].forward(flat, xa, None, ea, ha[1]), ha[0].loss(xa, 10.0, ha[1]), ha[0].backward(flat, xa, None, ea, g, ha[1], lambda_kl=1.0)))
th2 = timed(lambda: [(p.forward(flat, xa, None, ea, w), p.loss(xa, 10.0, w), p.backward(flat, xa, None, ea, g, w, lambda_kl=1.0)) for p, w in (ha, hb)])
print(f"(d) B=128 plan alone: {th:.3f} ms per pass; slot 0 then slot 1: {th2:.3f} ms per two passes")
