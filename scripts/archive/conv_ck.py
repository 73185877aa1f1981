"""Chunk-depth sweep of the k=5 conv at small T (GPU only): avc_set_tuning("conv_ck5", ck) per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["x"]
import conv_micro as m
for ck in (8, 16, 32):
    print(f"--- CK={ck}", flush=True)
    assert m.lib.avc_set_tuning(b"conv_ck5", ck) == 0
    for T in (128, 64, 32, 16):
        m.run(256, 128, 128, T, 5, 1, tiles=(11,), which="fd")
m.lib.avc_set_tuning(b"conv_ck5", 8)
