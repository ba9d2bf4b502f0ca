import sys, torch
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-configs", "--scaling", "weak", "--no-breakdown"]
from torch.profiler import profile, ProfilerActivity
import runpy
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    try:
        runpy.run_path("/root/repo/bench.py", run_name="__main__")
    except SystemExit:
        pass
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
