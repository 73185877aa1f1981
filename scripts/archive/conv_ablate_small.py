"""conv_ablate.py on the SMALL-GRID launches (the decoder chains of the B = 256 step run B = 128 halves at T_l = 16 / 32; the B = 4 step): where do the
~21 us of a launch go?  Variants: full | no LDS-DMA after the first chunks | + no barrier | no MFMA | empty loop | empty loop + no epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_ablate import run
for B, T in ((128, 16), (128, 32), (4, 128), (4, 16)):
    run(B, 128, 128, T, 5, (11,))
    run(B, 128, 128, T, 5, (11,), "d")
