"""HIP-graph replay of the train step with every kernel on ONE stream (a linear graph), small batches.  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from adaptive_voice_conversion_amd.config import default_config
from adaptive_voice_conversion_amd.solver import Solver
from adaptive_voice_conversion_amd import _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
def make():
    torch.manual_seed(0)
    return Solver(default_config(80), SimpleNamespace())
for single in (1, 0):
    lib.avc_set_single_stream(single)
    for B in (1, 4, 16, 64):
        T = 128
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, 80, T, generator=g).to(dev)
        eps = torch.randn(B, 128, T // 8, generator=g).to(dev)
        s = make()
        s.ae_step(x, 1.0, eps=eps, sync=True)
        n = 30
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): s.ae_step(x, 1.0, eps=eps, sync=False)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n * 1e3
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                s.ae_step(x, 1.0, eps=eps, sync=False)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3): graph.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): graph.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n * 1e3
        print(f"single_stream={single} B={B}: eager {te:.3f} ms/step, graph replay {tg:.3f} ms/step", flush=True)
        del graph, s
