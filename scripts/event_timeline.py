"""Per-stream timeline of ONE train step taken with HIP events around every launch (avc_prof_timeline) instead of a tracer: rocprofv3's
kernel trace stretches the step by ~10 % and may itself reorder what it observes.  Usage (GPU box):
    python scripts/event_timeline.py [--batch 256] [--tune name=value ...] [--presleep-ms 8] > timeline.txt"""
import argparse
import ctypes
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptive_voice_conversion_amd import _lib  # noqa: E402
from adaptive_voice_conversion_amd.config import default_config  # noqa: E402
from adaptive_voice_conversion_amd.solver import Solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--frames", type=int, default=128)
ap.add_argument("--tune", action="append", default=[])
ap.add_argument("--dtype", default="fp32", help="compute_dtype of the plan: fp32 | bf16s | bf16r | fp32x3")
ap.add_argument("--presleep-ms", type=float, default=8.0, help="GPU-side sleep in front of the recorded step: the host is then a whole step ahead")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
lib = _lib.load()
lib.avc_prof_timeline.restype = ctypes.c_int
lib.avc_prof_class_name.restype = ctypes.c_char_p
tuning = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.tune}
cfg = default_config(80)
if a.dtype != "fp32":
    cfg["compute_dtype"] = a.dtype
torch.manual_seed(0)
solver = Solver(cfg, types.SimpleNamespace(store_model_path=None, load_model=False, data_dir=None, logdir="/tmp/avc_tl_log", tuning=tuning))
x = torch.randn(a.batch, 80, a.frames, generator=torch.Generator().manual_seed(1)).to(dev)
eps = torch.randn(a.batch, 128, a.frames // 8, generator=torch.Generator().manual_seed(2)).to(dev)
for _ in range(5):
    solver.ae_step(x, 1.0, eps=eps, sync=False)
torch.cuda.synchronize()
# un-bracketed reference: time of a step without any event
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10):
    solver.ae_step(x, 1.0, eps=eps, sync=False)
t1.record()
torch.cuda.synchronize()
print(f"# un-bracketed step: {t0.elapsed_time(t1) / 10:.3f} ms")
if a.presleep_ms > 0:
    torch.cuda._sleep(int(a.presleep_ms * 2.0e6))
lib.avc_prof_begin()
solver.ae_step(x, 1.0, eps=eps, sync=False)
torch.cuda.synchronize()
N = 4096
cls, st = (ctypes.c_int * N)(), (ctypes.c_long * N)()
b, e = (ctypes.c_double * N)(), (ctypes.c_double * N)()
n = lib.avc_prof_timeline(cls, st, b, e, N)
streams = []
for i in range(min(n, N)):
    if st[i] not in streams:
        streams.append(st[i])
print(f"# {n} bracketed launches on {len(streams)} streams; span {max(e[i] for i in range(min(n, N))) * 1e3:.0f} us (each bracket adds ~2 us)")
rows = sorted(range(min(n, N)), key=lambda i: b[i])
for i in rows:
    print(f"{b[i] * 1e3:9.1f} +{(e[i] - b[i]) * 1e3:8.1f}  s{streams.index(st[i])}  {lib.avc_prof_class_name(cls[i]).decode()}")

# ---- a handful of named points of the backward pass, one event each (the schedule is NOT perturbed), median of 9 steps
import statistics
lib.avc_prof_marks_end.restype = ctypes.c_int
names = ["backward starts (main)", "dense-stack backward issued (side)", "speaker chain's first kernel issued (side)", "content chain done (main)",
         "speaker chain done (side)", "decoder weight gradients done (wgrad stream)", "all joined (main)", "-"]
acc = [[] for _ in names]
for _ in range(9):
    torch.cuda.synchronize()
    lib.avc_prof_marks_begin()
    solver.ae_step(x, 1.0, eps=eps, sync=False)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 8)()
    lib.avc_prof_marks_end(ms)
    for i in range(8):
        if ms[i] == ms[i]:
            acc[i].append(ms[i])
print("# marks of the backward pass, ms from its start (median of 9 un-bracketed steps):")
for i, nm in enumerate(names):
    if acc[i]:
        print(f"#   {statistics.median(acc[i]):7.3f}  {nm}")
