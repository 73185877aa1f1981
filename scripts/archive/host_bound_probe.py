"""Is the train step bound by the host's launch rate?  Per configuration: host time to ISSUE a step (perf_counter around the un-synchronised
loop), wall time per step, and the same step replayed from a captured HIP graph (no host launches at all).  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from adaptive_voice_conversion_amd.config import default_config
from adaptive_voice_conversion_amd.solver import Solver
dev = torch.device("cuda", 0)
for dtype in ("fp32", "bf16s"):
    for B in (4, 64, 256):
        cfg = default_config(80)
        cfg["compute_dtype"] = dtype
        torch.manual_seed(0)
        s = Solver(cfg, SimpleNamespace(store_model_path=None, load_model=False, data_dir=None, logdir="/tmp/avc_probe", tuning={}))
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, 80, 128, generator=g).to(dev)
        eps = torch.randn(B, 128, 16, generator=g).to(dev)
        for _ in range(5):
            s.ae_step(x, 1.0, eps=eps, sync=False)
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            s.ae_step(x, 1.0, eps=eps, sync=False)
        t_issue = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        t_wall = (time.perf_counter() - t0) / n * 1e3
        msg = f"{dtype} B={B}: host issue {t_issue:.3f} ms/step, wall {t_wall:.3f} ms/step"
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    s.ae_step(x, 1.0, eps=eps, sync=False)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                graph.replay()
            torch.cuda.synchronize()
            msg += f", HIP graph replay {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step"
            del graph
        except Exception as e:
            msg += f", graph capture failed: {repr(e)[:120]}"
        print(msg, flush=True)
        del s
