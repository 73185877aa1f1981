"""Effective shader clock of the conv kernel and its ablations (GPU only; run under
`rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace`; scripts/clock_probe_parse.py turns the CSVs into MHz = cycles / duration).
Every variant is preceded by a marker kernel (a torch elementwise add on 4 floats) so that the parser can split the dispatch sequence."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from adaptive_voice_conversion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
from conv_micro import pack

VARIANTS = ((0, "full"), (1, "noDMA"), (2, "noMFMA"), (7, "empty"))
marker = torch.zeros(4, device=dev)

def run(B, Cin, Cout, T, KS, tile, n=40, zero=False):
    x = torch.zeros(B, Cin, T, device=dev) if zero else torch.randn(B, Cin, T, device=dev)
    w = torch.randn(Cout, Cin, KS, device=dev) / (Cin * KS) ** 0.5
    if zero: w.zero_()
    b = torch.randn(Cout, device=dev)
    out = torch.zeros(B, Cout, T, device=dev)
    wp = pack(w, 0)
    for dbg, name in VARIANTS:
        lib.avc_set_tuning(b"conv_ablation", dbg)
        marker.add_(1.0)
        for _ in range(n):
            assert lib.avc_conv1d_fwd(P(x), x.stride(0), x.stride(1), 1, B, Cin, T, P(wp), P(b), Cout, KS, 1, 1, P(out),
                                      out.stride(0), out.stride(1), 1, 1, None, 0, 0, 0, 0, 0, None, tile, None) == 0
        torch.cuda.synchronize()
        print(f"group: B={B} {Cin}->{Cout} T={T} k={KS} t{tile} {'zeros ' if zero else ''}{name}", flush=True)
    lib.avc_set_tuning(b"conv_ablation", 0)

if __name__ == "__main__":
    run(256, 128, 128, 128, 5, 11)
    run(256, 128, 128, 128, 5, 11, zero=True)
    run(256, 1104, 128, 128, 1, 21)
    marker.add_(1.0)
    torch.cuda.synchronize()
