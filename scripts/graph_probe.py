"""Does the whole train step capture into a HIP graph (the C ABI claims to be capture safe)?  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from adaptive_voice_conversion_amd.config import default_config
from adaptive_voice_conversion_amd.solver import Solver
dev = torch.device("cuda", 0)
def make():
    torch.manual_seed(0)
    return Solver(default_config(80), SimpleNamespace())
for B in (256, 4):
    T = 128
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 80, T, generator=g).to(dev)
    eps = torch.randn(B, 128, T // 8, generator=g).to(dev)
    ref = make()
    for _ in range(4): m_ref = ref.ae_step(x, 1.0, eps=eps, sync=True)
    s = make()
    s.ae_step(x, 1.0, eps=eps, sync=True)           # eager warm-up (plan, workspace)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = s.ae_step(x, 1.0, eps=eps, sync=False)
    torch.cuda.current_stream().wait_stream(side)
    # the capture itself does not execute: steps 2..4 by replay
    for _ in range(3): graph.replay()
    torch.cuda.synchronize()
    m = {k: float(v) for k, v in out.items()}
    print(B, "eager ", m_ref)
    print(B, "graph ", m)
    pd = (ref.model.flat_parameters() - s.model.flat_parameters()).abs().max().item()
    print(B, "max |param diff| after 4 steps:", pd)
    n = 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): graph.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ref.ae_step(x, 1.0, eps=eps, sync=False)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B}: eager {te:.3f} ms/step, graph replay {tg:.3f} ms/step", flush=True)
