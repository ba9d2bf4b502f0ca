"""cProfile of the host side of the bench step loop at 10k points (where the GPU is not the limit): python tools/hostprof.py"""
import sys, cProfile, pstats, io
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "200", "--warmup", "5", "--points", "10000", "--no-cpu-baseline", "--no-configs", "--scaling", "weak", "--no-breakdown", "--no-layers"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
