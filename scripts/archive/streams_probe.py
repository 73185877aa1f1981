"""Does the number of launch plans ALIVE in the process (each owns 3 helper HIP streams + events) change the speed of a step that uses only one of them?"""
import copy, gc, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd.solver import Solver
from adaptive_voice_conversion_amd.engine import Plan
from bench import stock_config
dev = torch.device("cuda", 0)
cfg = stock_config(80)
torch.manual_seed(0)
s = Solver(copy.deepcopy(cfg), types.SimpleNamespace(store_model_path=None, load_model=False, data_dir=None, logdir="/tmp/avc_log", tuning={}))
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 128
x = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(1)).to(dev)
p = s.model._plan(B, T, T, dev)[0]
eps = torch.randn(B, 128, p.latent_len, generator=torch.Generator().manual_seed(2)).to(dev)

def timed(n=30, w=5):
    for _ in range(w): s.ae_step(x, 1.0, eps=eps, sync=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): s.ae_step(x, 1.0, eps=eps, sync=False)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n

print(f"B={B}: {timed():.3f} ms/step with 1 plan alive")
extra = []
for n in (1, 2, 4, 8, 16):
    while len(extra) < n:
        extra.append(Plan(cfg, 1, 64 + 8 * len(extra), mode="inference", device=dev))   # created, never launched
    print(f"   {timed():.3f} ms/step with {n} more plans alive (never launched)")
xs = torch.randn(1, 80, 64, device=dev)
for q in extra[:4]:   # launch four of them once
    ws = torch.zeros(q.workspace_floats, device=dev)
    xq = torch.randn(1, 80, q.T, device=dev)
    q.forward(s.model.flat_parameters(), xq, xq, None, ws)
torch.cuda.synchronize()
print(f"   {timed():.3f} ms/step after four of them ran one forward each")
for q in extra: q.close()
extra.clear(); gc.collect()
print(f"   {timed():.3f} ms/step after closing them all")
