"""BASELINE cfg3 with bf16 against f32 feature rows in the depth-wise layers: step time and per-layer forward / backward."""
import os, sys, types, torch
sys.path.insert(0, os.getcwd())
import bench
from mccnn_amd import workloads as W
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
args = types.SimpleNamespace(no_pipeline=False)
res = {}
for bf in (True, False):
    c = W.CONFIGS["cfg3"]
    W.CONFIGS["cfg3"] = c._replace(convs=W.mcseg(32, bf16=bf))
    e = bench.run_config("cfg3", torch.device("cuda", 0), args, False)
    print("bf16 rows", bf, e["mode"], e["ms_per_step"], "seq", e["sequential_ms_per_step"], flush=True)
    res[bf] = e["layers"]
for a, b in zip(res[True], res[False]):
    print("%-9s fin %3d edges %8d  fwd bf16 %.4f f32 %.4f   bwd bf16 %.4f f32 %.4f" % (
        a["name"], a["fin"], a["edges"], a["fwd_ms"], b["fwd_ms"], a["bwd_ms"], b["bwd_ms"]))
