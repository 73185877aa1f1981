"""(record of a removed experiment: the conv_stagger knob is no longer in the library, profiles/r03_stagger.log)
conv micro-benchmark at the model's heavy shapes with avc_tuning.conv_stagger = argv[1] (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_micro as cm
cm.lib.avc_set_tuning(b"conv_stagger", int(sys.argv[1]))
for (B, Cin, Cout, T, KS, s, tiles) in ((256, 128, 128, 128, 5, 1, (11,)), (256, 128, 128, 64, 5, 1, (11,)), (256, 128, 128, 32, 5, 1, (11,)),
                                        (256, 1104, 128, 128, 1, 1, (21,)), (256, 80, 128, 128, 8, 1, (11,))):
    cm.run(B, Cin, Cout, T, KS, s, tiles, "fd" if KS != 8 else "f")
