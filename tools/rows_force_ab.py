"""Per-layer forward / backward times of a BASELINE configuration with the plans and geometry cached (bench.ConfigWorkload.per_layer):
run once as it is and once with MCCNN_DEBUG=rows_force=1 to re-check the rows / streaming rule of exec.hip:rows_shape at HEAD.
    python tools/rows_force_ab.py cfg3 [cfg2 ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mccnn_amd.workloads import CONFIGS  # noqa: E402

torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
for name in sys.argv[1:] or ["cfg3"]:
    cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
    _, layers, _ = cw.per_layer(iters=8)
    for a in layers:
        if a["combin"]:
            continue
        print("%s %-9s F=%-4d pts %-7d centres %-7d e %-8d fwd %.3f bwd %.3f" % (
            name, a["name"], a["fin"], a["points_in"], a["centres"], a["edges"], a["fwd_ms"], a["bwd_ms"]))
