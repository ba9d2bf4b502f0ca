"""Per-layer forward / backward HIP-event times (geometry cached) of a BASELINE configuration: python tools/layer_table.py cfg4"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from mccnn_amd.workloads import CONFIGS
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
for name in sys.argv[1:]:
    cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
    t_h, layers, sizes = cw.per_layer(5)
    print("%s: hierarchy %.3f ms; sum fwd %.3f bwd %.3f" % (name, t_h, sum(l["fwd_ms"] for l in layers), sum(l["bwd_ms"] for l in layers)))
    for l in layers:
        print("  %-9s lv %s fin %4d %s n %6d m %6d E %8d nb %3d  fwd %.3f bwd %.3f" % (l["name"], l["levels"], l["fin"], "C" if l["combin"] else "D",
              l["points_in"], l["centres"], l["edges"], l["mlp_blocks"], l["fwd_ms"], l["bwd_ms"]))
