"""Micro-benchmark of the InstanceNorm/AdaIN row kernels (back-to-back launches between two events)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adaptive_voice_conversion_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, C = 256, 128
for T in (128, 64, 32, 16):
    # rotate over several buffers so that the working set (8 x 3 tensors) exceeds the 256 MiB Infinity Cache at T=128
    NB = 8
    ys = [torch.randn(B, C, T, device=dev) for _ in range(NB)]; outs = [torch.empty_like(y) for y in ys]; gs = [torch.randn_like(y) for y in ys]
    dys = [torch.empty_like(y) for y in ys]
    cond = torch.randn(B, 2 * C, device=dev); mean = torch.empty(B * C, device=dev); rstd = torch.empty(B * C, device=dev); dcond = torch.zeros(B, 2 * C, device=dev)
    i = [0]
    def f():
        k = i[0] % NB; i[0] += 1
        lib.avc_instnorm_fwd(P(ys[k]), B, C, T, P(cond), 2 * C, 0, 1, None, 0, 0, P(outs[k]), P(mean), P(rstd), None)
    def b():
        k = i[0] % NB; i[0] += 1
        lib.avc_instnorm_bwd(P(gs[k]), P(ys[k]), P(mean), P(rstd), B, C, T, P(cond), 2 * C, 0, 1, P(dys[k]), P(dcond), 2 * C, 0, None)
    f(); tf = timeit(f); tb = timeit(b)
    n = B * C * T * 4
    print(f"T={T:4d}: fwd {tf:6.2f}us {2*n/tf/1e3:7.1f} GB/s | bwd {tb:6.2f}us {3*n/tb/1e3:7.1f} GB/s", flush=True)
