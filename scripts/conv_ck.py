"""Chunk-depth sweep of the k=5 conv at small T (GPU only): AVC_CONV_CK5 is read once per process."""
import subprocess, sys, os
code = '''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath("scripts/conv_micro.py")))
sys.argv = ["x"]
import conv_micro as m
for T in (128, 64, 32, 16):
    m.run(256, 128, 128, T, 5, 1, tiles=(11,), which="fd")
'''
for ck in (8, 16, 32):
    print(f"--- CK={ck}", flush=True)
    env = dict(os.environ, AVC_CONV_CK5=str(ck))
    subprocess.run([sys.executable, "-c", code], env=env)
