"""Sequential against pipelined steps of the BASELINE configurations (bench.run_config: the next batch's PointHierarchy
requested one step ahead): python tools/config_pipe.py [cfgN ...]"""
import os, sys, types, torch
sys.path.insert(0, os.getcwd())
import bench
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
args = types.SimpleNamespace(no_pipeline=False)
for name in (sys.argv[1:] or ["cfg0", "cfg1", "cfg2", "cfg3", "cfg4"]):
    e = bench.run_config(name, torch.device("cuda", 0), args, False)
    print("%s: %s %.3f ms/step (sequential %.3f), host issue %.3f, own %.3f, %.0f launches" % (
        name, e["mode"], e["ms_per_step"], e["sequential_ms_per_step"], e["host_issue_ms_per_step"], e["host_busy_ms_per_step"],
        e["library_launches_per_step"]), flush=True)
