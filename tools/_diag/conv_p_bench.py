"""The producer / consumer conv (KA_CONV_P=1) against conv3x3_kernel on the four launch kinds: bit-identity and stand-alone time."""
import os, sys
sys.path.insert(0, '.')
os.environ["CG_VAR"] = "KA_CONV_P"
os.environ.setdefault("CG_ON", "3")
os.environ.setdefault("CG_KINDS", "0,1,2,3")
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["4096"])
exec(open("tools/conv_g_bench.py").read())
