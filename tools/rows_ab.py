"""Row-per-lane against edge-streaming kernels, layer by layer of a BASELINE configuration (HIP-event times with the
geometry and the plans cached): python tools/rows_ab.py cfg3 [cfg4 ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mccnn_amd import MCConvModule as M  # noqa: E402
from mccnn_amd.workloads import CONFIGS  # noqa: E402

torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
for name in sys.argv[1:] or ["cfg3"]:
    res = {}
    hint = M._order_hint
    for rows in (True, False, "nohint"):
        M.ROW_KERNELS = bool(rows)
        M._order_hint = (lambda p: None) if rows == "nohint" else hint
        cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
        _, layers, _ = cw.per_layer(iters=8)
        res[rows] = layers
    M.ROW_KERNELS, M._order_hint = True, hint
    print("== %s: rows fwd / bwd | streaming fwd / bwd | rows without visiting-order hints fwd / bwd (ms)" % name)
    for a, b, c in zip(res[True], res[False], res["nohint"]):
        if a["combin"]:
            continue
        print("%-9s F=%-4d pts %-7d centres %-7d e %-8d | %.3f %.3f | %.3f %.3f | %.3f %.3f" % (
            a["name"], a["fin"], a["points_in"], a["centres"], a["edges"], a["fwd_ms"], a["bwd_ms"],
            b["fwd_ms"], b["bwd_ms"], c["fwd_ms"], c["bwd_ms"]))
