import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_micro as m
which = sys.argv[1] if len(sys.argv) > 1 else "f"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 11
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
m.run(256, 128, 128, T, 5, 1, tiles=(tile,) if which != "w" else (), which=which)
