import ctypes, os, sys, json, subprocess
os.environ["MCCNN_LIB_NAME"] = "lib_timing.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-breakdown"] + sys.argv[1:]
import runpy, io, contextlib
from mccnn_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
import torch
runpy.run_path("bench.py", run_name="__main__")
torch.cuda.synchronize()
lib.mccnn_debug_timing(buf, 0)
v = list(buf)
tot = sum(v)
names = ["-", "loads issued", "MLP L1-L3 + masks", "dfeat + gf", "dW3 FMAs", "t3 MFMA+mask", "dW2 FMAs", "t4 MFMA", "dW1 FMAs"]
for k in range(1, 9):
    print("%-22s %12d cycles  %5.1f%%" % (names[k], v[k], 100.0 * v[k] / max(tot, 1)))
