"""Round 5: tile walk / scheduling knobs of conv_gemm at model shapes, op level (GPU only).  usage: walk_micro.py [quick]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import conv_micro as m
lib = m.lib

def tune(**kw):
    for k, v in kw.items():
        assert lib.avc_set_tuning(k.encode(), v) == 0, k

SHAPES = [  # B, Cin, Cout, T, KS, stride, which
    (256, 128, 128, 128, 5, 1, "fd"),
    (256, 128, 128, 64, 5, 1, "fd"),
    (256, 128, 128, 128, 5, 2, "fd"),
    (256, 80, 128, 128, 8, 1, "f"),
    (1024, 80, 128, 128, 8, 1, "f"),
    (256, 1104, 128, 128, 1, 1, "fd"),
    (1024, 128, 128, 128, 5, 1, "fd"),
    (64, 128, 128, 1024, 5, 1, "fd"),
    (256, 128, 256, 64, 5, 1, "f"),
]
VARIANTS = [
    ("warm-up (discard)", dict(conv_walk=0, conv_sched=0)),
    ("base", dict(conv_walk=0, conv_sched=0)),
    ("prio static", dict(conv_walk=0, conv_sched=1)),
    ("prio rotate", dict(conv_walk=0, conv_sched=2)),
    ("walk4", dict(conv_walk=4, conv_sched=0)),
    ("walk3", dict(conv_walk=3, conv_sched=0)),
    ("walk2", dict(conv_walk=2, conv_sched=0)),
    ("walk2 stagger8", dict(conv_walk=2, conv_sched=8 << 8)),
    ("walk4 rotate", dict(conv_walk=4, conv_sched=2)),
    ("walk4 static", dict(conv_walk=4, conv_sched=1)),
    ("base again", dict(conv_walk=0, conv_sched=0)),
]
for name, kw in VARIANTS:
    print("##", name, kw, flush=True)
    tune(**kw)
    for (B, Ci, Co, T, KS, s, which) in SHAPES:
        m.run(B, Ci, Co, T, KS, s, tiles=(0,), which=which)
tune(conv_walk=0, conv_sched=0)
