"""Train-step rate when every batch starts in host memory (pageable / pinned) instead of HBM (GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from adaptive_voice_conversion_amd.config import default_config
from adaptive_voice_conversion_amd.solver import Solver
dev = torch.device("cuda", 0)
torch.manual_seed(0)
s = Solver(default_config(80), SimpleNamespace())
B, T = 256, 128
eps = torch.randn(B, 128, T // 8, device=dev)
def run(x, n=20):
    for _ in range(3): s.ae_step(x, 1.0, eps=eps, sync=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): s.ae_step(x, 1.0, eps=eps, sync=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
xh = torch.randn(B, T, 80)                      # collate layout [B,T,M] (data_utils.py:14-16), viewed [B,M,T]
print(f"resident in HBM : {run(xh.to(dev).transpose(1, 2)):.3f} ms/step")
print(f"pageable host   : {run(xh.transpose(1, 2)):.3f} ms/step")
print(f"pinned host     : {run(xh.pin_memory().transpose(1, 2)):.3f} ms/step")
