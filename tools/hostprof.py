import sys, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--steps", "200", "--warmup", "5", "--points", "10000", "--no-cpu-baseline", "--no-breakdown"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
